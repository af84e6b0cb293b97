"""ctypes binding of oracle/libbpe_oracle.so -- TEST INFRASTRUCTURE (the checker, never the product)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(ORACLE_DIR, "libbpe_oracle.so")
        src = os.path.join(ORACLE_DIR, "bpe_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["make", "-C", ORACLE_DIR, "oracle"], check=True, capture_output=True)
        L = C.CDLL(so)
        L.oracle_train.restype = C.c_int
        L.oracle_train.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_char_p, C.c_char_p, C.c_int]
        L.oracle_char_hist.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(u32p), C.POINTER(u64p), u64p, u64p]
        L.oracle_alphabet.argtypes = [u32p, u64p, C.c_uint64, C.c_uint64, C.c_double, C.c_int, C.POINTER(u32p),
                                      C.POINTER(u32p), u64p, C.POINTER(u32p), u64p]
        L.oracle_word_table.argtypes = [C.c_char_p, C.c_uint64, u32p, u32p, C.c_uint64, C.c_uint32, C.POINTER(u32p),
                                        C.POINTER(u64p), C.POINTER(u64p), u64p]
        L.oracle_pair_counts.argtypes = [u32p, u64p, u64p, C.c_uint64, C.POINTER(u32p), C.POINTER(u32p),
                                         C.POINTER(u64p), u64p]
        L.oracle_apply_rules.argtypes = [u32p, u64p, C.c_uint64, u32p, C.c_uint64]
        L.oracle_learn_rules.argtypes = [u32p, u64p, u64p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(u32p),
                                         C.POINTER(u64p), u64p]
        L.oracle_ska_order.argtypes = [u32p, C.c_uint64, u32p]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_model_load.restype = C.c_void_p
        L.oracle_model_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.oracle_model_free.argtypes = [C.c_void_p]
        L.oracle_model_vocab_size.argtypes = [C.c_void_p]
        L.oracle_encode.restype = C.c_int64
        L.oracle_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_double, i32p,
                                    C.c_uint64, C.c_char_p, C.c_int]
        L.oracle_encode_batch.argtypes = [C.c_void_p, C.c_char_p, u64p, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.POINTER(i32p), C.POINTER(u64p), C.c_char_p, C.c_int]
        _LIB = L
    return _LIB


def _take(ptr, n, dtype):
    n = int(n)
    if n == 0:
        arr = np.zeros(0, dtype=dtype)
    else:
        arr = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    lib().oracle_free(C.cast(ptr, C.c_void_p))
    return arr


def _p(a, t):
    return a.ctypes.data_as(t)


def train(text: bytes, model_path: str, vocab_size: int, coverage=1.0, pad_id=0, unk_id=1, bos_id=2, eos_id=3):
    err = C.create_string_buffer(1024)
    rc = lib().oracle_train(text, len(text), vocab_size, coverage, pad_id, unk_id, bos_id, eos_id,
                            model_path.encode(), err, 1024)
    if rc != 0:
        raise ValueError(err.value.decode())


def char_hist(text: bytes):
    cps, cnts = u32p(), u64p()
    n, dl = C.c_uint64(), C.c_uint64()
    lib().oracle_char_hist(text, len(text), C.byref(cps), C.byref(cnts), C.byref(n), C.byref(dl))
    return _take(cps, n.value, np.uint32), _take(cnts, n.value, np.uint64), dl.value


def alphabet(cps, cnts, data_len, coverage, n_special):
    cps = np.ascontiguousarray(cps, np.uint32)
    cnts = np.ascontiguousarray(cnts, np.uint64)
    ocp, oid, orm = u32p(), u32p(), u32p()
    n, nr = C.c_uint64(), C.c_uint64()
    lib().oracle_alphabet(_p(cps, u32p), _p(cnts, u64p), len(cps), data_len, coverage, n_special, C.byref(ocp),
                          C.byref(oid), C.byref(n), C.byref(orm), C.byref(nr))
    return _take(ocp, n.value, np.uint32), _take(oid, n.value, np.uint32), _take(orm, nr.value, np.uint32)


def word_table(text: bytes, cp_map, id_map, space_id):
    cp_map = np.ascontiguousarray(cp_map, np.uint32)
    id_map = np.ascontiguousarray(id_map, np.uint32)
    tok, off, cnt = u32p(), u64p(), u64p()
    n = C.c_uint64()
    lib().oracle_word_table(text, len(text), _p(cp_map, u32p), _p(id_map, u32p), len(cp_map), space_id,
                            C.byref(tok), C.byref(off), C.byref(cnt), C.byref(n))
    U = n.value
    off_a = _take(off, U + 1, np.uint64)
    T = int(off_a[-1]) if U else 0
    return _take(tok, T, np.uint32), off_a, _take(cnt, U, np.uint64)


def pair_counts(tok, off, cnt):
    tok = np.ascontiguousarray(tok, np.uint32)
    off = np.ascontiguousarray(off, np.uint64)
    cnt = np.ascontiguousarray(cnt, np.uint64)
    xs, ys, cs = u32p(), u32p(), u64p()
    n = C.c_uint64()
    lib().oracle_pair_counts(_p(tok, u32p), _p(off, u64p), _p(cnt, u64p), len(cnt), C.byref(xs), C.byref(ys),
                             C.byref(cs), C.byref(n))
    return _take(xs, n.value, np.uint32), _take(ys, n.value, np.uint32), _take(cs, n.value, np.uint64)


def apply_rules(tok, off, rules_xyz):
    tok = np.array(tok, np.uint32, copy=True)
    off = np.array(off, np.uint64, copy=True)
    r = np.ascontiguousarray(rules_xyz, np.uint32).reshape(-1)
    lib().oracle_apply_rules(_p(tok, u32p), _p(off, u64p), len(off) - 1, _p(r, u32p), len(r) // 3)
    return tok[: int(off[-1])], off


def learn_rules(tok, off, cnt, first_new_id, max_rules):
    tok = np.ascontiguousarray(tok, np.uint32)
    off = np.ascontiguousarray(off, np.uint64)
    cnt = np.ascontiguousarray(cnt, np.uint64)
    rules, rc = u32p(), u64p()
    n = C.c_uint64()
    ret = lib().oracle_learn_rules(_p(tok, u32p), _p(off, u64p), _p(cnt, u64p), len(cnt), first_new_id, max_rules,
                                   C.byref(rules), C.byref(rc), C.byref(n))
    assert ret == 0
    r = _take(rules, 3 * n.value, np.uint32).reshape(-1, 3)
    c = _take(rc, max(n.value, 0), np.uint64)
    return r, c


def ska_order(keys):
    keys = np.ascontiguousarray(keys, np.uint32)
    out = np.zeros(len(keys), np.uint32)
    rc = lib().oracle_ska_order(_p(keys, u32p), len(keys), _p(out, u32p))
    assert rc == 0
    return out


class Model:
    def __init__(self, path):
        err = C.create_string_buffer(1024)
        self.h = lib().oracle_model_load(path.encode(), err, 1024)
        if not self.h:
            raise ValueError(err.value.decode())

    def __del__(self):
        if getattr(self, "h", None):
            lib().oracle_model_free(self.h)
            self.h = None

    def vocab_size(self):
        return lib().oracle_model_vocab_size(self.h)

    def encode(self, sentences, bos=False, eos=False, reverse=False, dropout_prob=0.0):
        """sentences: list[bytes] -> list[list[int]]"""
        ids, off = self.encode_packed(sentences, bos, eos, reverse, dropout_prob)
        return [ids[int(off[i]):int(off[i + 1])].tolist() for i in range(len(sentences))]

    def encode_packed(self, sentences, bos=False, eos=False, reverse=False, dropout_prob=0.0):
        blob = b"".join(sentences)
        offs = np.zeros(len(sentences) + 1, np.uint64)
        np.cumsum([len(s) for s in sentences], out=offs[1:])
        return self.encode_blob(blob, offs, bos, eos, reverse, dropout_prob)

    def encode_blob(self, blob, offs, bos=False, eos=False, reverse=False, dropout_prob=0.0):
        offs = np.ascontiguousarray(offs, np.uint64)
        ids, ooff = i32p(), u64p()
        err = C.create_string_buffer(1024)
        rc = lib().oracle_encode_batch(self.h, bytes(blob), _p(offs, u64p), len(offs) - 1, int(bos), int(eos),
                                       int(reverse), float(dropout_prob), C.byref(ids), C.byref(ooff), err, 1024)
        if rc != 0:
            raise ValueError(err.value.decode())
        off = _take(ooff, len(offs), np.uint64)
        return _take(ids, int(off[-1]), np.int32), off


def rng_reset():
    lib().oracle_rng_reset()
