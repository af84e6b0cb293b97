import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# A real MI355X is visible iff the KFD device node exists.  Without one, the kernel-logic tests run the UNMODIFIED
# product sources against the HIP emulator in tests/hipsim (test infrastructure; never a product fallback).
HAVE_GPU = os.path.exists("/dev/kfd")
SIM_LIB = os.path.join(ROOT, "tests", "hipsim", "_build", "libyttm_sim.so")
if not HAVE_GPU and "YTTM_AMD_LIB" not in os.environ:
    os.environ["YTTM_AMD_LIB"] = SIM_LIB
if not HAVE_GPU:
    # the emulator's time goes with the workgroups it runs: word-mode rounds sized for a few thousand words instead of the 16 k the
    # MI355X is given (a grid size only -- the kernels are the same)
    os.environ.setdefault("YTTM_WORD_HINT_FLOOR", "2048")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the reference compiled from /root/reference)")


@pytest.fixture(scope="session")
def sim_lib():
    """Builds tests/hipsim/_build/libyttm_sim.so (g++, product sources + emulator).  Skips when a real GPU is used."""
    if HAVE_GPU:
        pytest.skip("real GPU present: the emulator build is only used on GPU-less machines")
    import fcntl
    with open(os.path.join(ROOT, "tests", "hipsim", ".build.lock"), "w") as lock:  # (pytest-xdist: one worker builds, the others wait)
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "hipsim"), "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("hipsim build failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
    return SIM_LIB
