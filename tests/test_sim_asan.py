"""The product's kernels under the HIP emulator AND AddressSanitizer: a wrong index that the GPU would turn into a memory
fault (or silently read) is reported on the CPU with file and line."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _libasan():
    try:
        p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    except OSError:
        return None
    return p if p and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_libasan() is None, reason="libasan not available")
def test_kernels_under_address_sanitizer():
    r = subprocess.run(["make", "-C", os.path.join(HERE, "hipsim"), "-j8", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    env = dict(os.environ)
    env["LD_PRELOAD"] = _libasan()
    env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1"
    env["YTTM_AMD_LIB"] = os.path.join(HERE, "hipsim", "_build_asan", "libyttm_sim_asan.so")
    r = subprocess.run([sys.executable, os.path.join(HERE, "asan_scenarios.py")], capture_output=True, text=True, env=env, timeout=900)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and "ASAN_SCENARIOS_OK" in r.stdout, tail
