// yttm_config.cpp -- the one place that reads YTTM_* environment variables (yttm_config.h).
#include "yttm_config.h"

#include <stdlib.h>

#include <atomic>
#include <mutex>

namespace yttm {
namespace {

void read_hook(Hook &h, const char *name, const char *dflt) {
  const char *v = getenv(name);
  h.set = v != nullptr;
  h.raw = v ? v : "";
  const char *src = (v && *v) ? v : dflt;  // an empty value counts as "not given" for the numbers (as env_uint always did)
  h.u = *src ? strtoull(src, nullptr, 10) : 0ull;
  h.i = *src ? strtoll(src, nullptr, 10) : 0ll;
  h.d = *src ? atof(src) : 0.0;
}

std::mutex g_mu;
std::shared_ptr<const Config> g_cfg;

std::shared_ptr<const Config> make() {
  auto c = std::make_shared<Config>();
#define X(field, name, dflt, kind, doc) read_hook(c->field, name, dflt);
  YTTM_HOOKS(X)
#undef X
  return c;
}

thread_local std::shared_ptr<const Config> tl_bound;

}  // namespace

CfgBind::CfgBind(std::shared_ptr<const Config> c) : bound_(c != nullptr) {
  if (!bound_) return;
  prev_ = std::move(tl_bound);
  tl_bound = std::move(c);
}
CfgBind::~CfgBind() {
  if (bound_) tl_bound = std::move(prev_);
}

std::shared_ptr<const Config> cfg() {
  if (tl_bound) return tl_bound;
  std::lock_guard<std::mutex> g(g_mu);
  if (!g_cfg) g_cfg = make();
  return g_cfg;
}

void cfg_refresh() {
  std::shared_ptr<const Config> fresh = make();
  std::lock_guard<std::mutex> g(g_mu);
  g_cfg = std::move(fresh);
}

const char *config_table_markdown() {
  static const std::string table = [] {
    std::string t = "| variable | default | kind | what it does |\n|---|---|---|---|\n";
#define X(field, name, dflt, kind, doc) t += std::string("| `") + name + "` | " + (*dflt ? dflt : "unset") + " | " + kind + " | " + doc + " |\n";
    YTTM_HOOKS(X)
#undef X
    return t;
  }();
  return table.c_str();
}

}  // namespace yttm
