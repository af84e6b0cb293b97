"""run_select.h (the mask arithmetic of the position-parallel merge-apply kernel) against brute force: plain C++, no GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_select_brute_force(tmp_path):
    exe = str(tmp_path / "run_select_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "youtokentome_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "run_select_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
