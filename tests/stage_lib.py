"""ctypes wrapper over the inner C ABI (include/yttm_gpu.h) used by the per-kernel parity checks."""
import ctypes as C

import numpy as np

from youtokentome_amd import _lib


class Ctx:
    def __init__(self, device=0):
        self.L = _lib.load()
        self.h = C.c_void_p()
        self._chk(self.L.yttm_gpu_ctx_create(device, C.byref(self.h)))
        self.n_unique = self.n_tokens = 0

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.yttm_gpu_last_error().decode())

    def close(self):
        if self.h:
            self.L.yttm_gpu_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def upload(self, text: bytes):
        self._chk(self.L.yttm_gpu_upload_corpus(self.h, text, len(text)))

    def char_hist(self):
        cap = 0x110000
        cps = np.zeros(cap, np.uint32)
        cnts = np.zeros(cap, np.uint64)
        n = C.c_uint32(cap)
        steps = C.c_uint64()
        self._chk(self.L.yttm_gpu_char_hist(self.h, cps.ctypes.data_as(_lib.u32p), cnts.ctypes.data_as(_lib.u64p), C.byref(n),
                                            C.byref(steps)))
        k = n.value
        order = np.argsort(cps[:k])
        return cps[:k][order], cnts[:k][order], steps.value

    def build_word_table(self, cp, ids, space_id, n_ids_cap):
        cp = np.ascontiguousarray(cp, np.uint32)
        ids = np.ascontiguousarray(ids, np.uint32)
        nu, nt = C.c_uint64(), C.c_uint64()
        self._chk(self.L.yttm_gpu_build_word_table(self.h, cp.ctypes.data_as(_lib.u32p), ids.ctypes.data_as(_lib.u32p), len(cp),
                                                   space_id, n_ids_cap, C.byref(nu), C.byref(nt)))
        self.n_unique, self.n_tokens = nu.value, nt.value
        return nu.value, nt.value

    def word_table(self):
        tok = np.zeros(max(self.n_tokens, 1), np.uint32)
        off = np.zeros(self.n_unique + 2, np.uint64)
        cnt = np.zeros(max(self.n_unique, 1), np.uint32)
        now = C.c_uint64()
        self._chk(self.L.yttm_gpu_download_word_table(self.h, tok.ctypes.data_as(_lib.u32p), off.ctypes.data_as(_lib.u64p),
                                                      cnt.ctypes.data_as(_lib.u32p), C.byref(now)))
        return tok[: now.value], off[: self.n_unique + 1], cnt[: self.n_unique]

    def words_as_multiset(self):
        tok, off, cnt = self.word_table()
        return sorted((tuple(tok[int(off[i]):int(off[i + 1])].tolist()), int(cnt[i])) for i in range(len(cnt)))

    def pair_count(self):
        n = C.c_uint64()
        self._chk(self.L.yttm_gpu_pair_count(self.h, C.byref(n)))
        return n.value

    def pairs(self, cap=1 << 20):
        keys = np.zeros(cap, np.uint64)
        cnts = np.zeros(cap, np.uint64)
        n = C.c_uint64(cap)
        self._chk(self.L.yttm_gpu_download_pairs(self.h, keys.ctypes.data_as(_lib.u64p), cnts.ctypes.data_as(_lib.u64p), C.byref(n)))
        k = n.value
        order = np.argsort(keys[:k])
        return keys[:k][order], cnts[:k][order]

    def merge_apply(self, rules_xyz):
        r = np.ascontiguousarray(rules_xyz, np.uint32).reshape(-1)
        self._chk(self.L.yttm_gpu_merge_apply(self.h, r.ctypes.data_as(_lib.u32p), len(r) // 3))

    def k4_measure(self, on=True, read=False):
        out = np.zeros(6, np.uint64)
        self._chk(self.L.yttm_gpu_k4_measure(self.h, int(on), out.ctypes.data_as(_lib.u64p) if read else None))
        return dict(zip(("sites", "tiles", "tile_tokens", "words", "word_tokens", "rounds"), (int(v) for v in out)))

    def pair_query(self, keys):
        keys = np.ascontiguousarray(keys, np.uint64)
        out = np.zeros(len(keys), np.uint64)
        self._chk(self.L.yttm_gpu_pair_query(self.h, keys.ctypes.data_as(_lib.u64p), len(keys), out.ctypes.data_as(_lib.u64p)))
        return out

    def candidates(self, tau_cnt, tau_mx=0xFFFFFFFF, cap=1 << 16):
        keys = np.zeros(cap, np.uint64)
        cnts = np.zeros(cap, np.uint64)
        n = C.c_uint32(cap)
        self._chk(self.L.yttm_gpu_candidates(self.h, tau_cnt, tau_mx, keys.ctypes.data_as(_lib.u64p), cnts.ctypes.data_as(_lib.u64p),
                                             C.byref(n)))
        k = min(n.value, cap)
        return keys[:k], cnts[:k], n.value
