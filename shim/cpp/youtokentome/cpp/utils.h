#pragma once
#include "../../vkcom_adapter.h"  // the reference's utils.h, as an adapter over the C ABI of libyttm_mi355x.so
