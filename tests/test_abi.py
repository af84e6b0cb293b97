"""The C-ABI shared library loads and exports every symbol that include/*.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "youtokentome_amd", "libyttm_mi355x.so")


def declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(yttm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "youtokentome_amd", "csrc"), "-j8"], check=True, capture_output=True)
    lib = ctypes.CDLL(LIB)
    names = declared("yttm_mi355x.h") + declared("yttm_gpu.h")
    assert len(names) > 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_lists_the_same_symbols():
    from youtokentome_amd import _lib
    names = set(declared("yttm_mi355x.h") + declared("yttm_gpu.h"))
    assert names <= set(_lib.EXPORTS) | {"yttm_comm_rccl_unique_id", "yttm_comm_rccl_create", "yttm_comm_destroy",
                                         "yttm_train_bpe_from_device_comm", "yttm_comm_callback_create"}


def test_no_gpu_means_loud_failure():
    """Without a GPU the product must fail loudly, never fall back to a CPU path."""
    if os.path.exists("/dev/kfd"):
        return
    code = ("import os,sys; os.environ.pop('YTTM_AMD_LIB',None); sys.path.insert(0,%r); import youtokentome_amd as y;\n"
            "open('/tmp/_abi_t.txt','w').write('aa bb ab')\n"
            "try:\n    y.BPE.train('/tmp/_abi_t.txt','/tmp/_abi_t.model',20)\n    print('TRAINED')\nexcept ValueError as e:\n    print('ERR', e)\n" % ROOT)
    r = subprocess.run(["python", "-c", code], capture_output=True, text=True)
    assert "TRAINED" not in r.stdout
    assert "ERR" in r.stdout and "GPU" in r.stdout


def test_python_list_boundary_extension_is_built():
    """csrc/pyapi.c (the C side of BPE.encode(list[str]) -> list[list[int]], yttm.pyx:87-109) is built next to the library and imports."""
    import sysconfig
    so = os.path.join(ROOT, "youtokentome_amd", "_yttm_pyapi" + sysconfig.get_config_var("EXT_SUFFIX"))  # (named with this interpreter's ABI tag)
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(ROOT, "youtokentome_amd", "csrc"), "pyapi", "PYTHON=" + sys.executable], check=True, capture_output=True)
    from youtokentome_amd import bpe
    assert bpe._pyapi is not None and hasattr(bpe._pyapi, "encode_ids")


def test_environment_hooks_are_one_table():
    """Every YTTM_* hook of the library lives in csrc/yttm_config.h: no other getenv in the sources, and INTEGRATION.md prints the table."""
    src = os.path.join(ROOT, "youtokentome_amd", "csrc")
    offenders = []
    for f in sorted(os.listdir(src)):
        if f.endswith((".cpp", ".hip", ".h", ".c")) and f != "yttm_config.cpp":
            for i, line in enumerate(open(os.path.join(src, f), errors="replace"), 1):
                if re.search(r"\bgetenv\s*\(", line) and not line.lstrip().startswith("//"):
                    offenders.append("%s:%d" % (f, i))
    assert not offenders, offenders
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", src, "-j8"], check=True, capture_output=True)
    lib = ctypes.CDLL(LIB)
    lib.yttm_config_table.restype = ctypes.c_char_p
    table = lib.yttm_config_table().decode()
    rows = [r for r in table.splitlines() if r.startswith("| `YTTM_")]
    assert len(rows) >= 70
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [r for r in rows if r not in doc]
    assert not missing, missing[:3]
    # every YTTM_ name a test or a tool sets must be a row (or one of the few variables read outside the library)
    known = set(re.findall(r"`(YTTM_[A-Z0-9_]+)`", table)) | {"YTTM_AMD_LIB", "YTTM_BENCH_FORCE_COMM", "YTTM_FULL_PINS", "YTTM_RUN_REFSUITE_ON_SIM", "YTTM_K4_PROF"}
    used = set()
    for d in ("tests", "tools", "."):
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            if "_ref" in base or "gpurun_out" in base or ".git" in base or (d == "." and base != ROOT):
                continue
            for f in files:
                if f.endswith((".py", ".sh")):
                    used |= set(re.findall(r"\b(YTTM_[A-Z0-9_]+)\b", open(os.path.join(base, f), errors="replace").read()))
    unknown = sorted(n for n in used - known if not n.startswith(("YTTM_TEST_EXPECT", "YTTM_TEST_FREE_BYTES_RANK")))
    assert not unknown, unknown
