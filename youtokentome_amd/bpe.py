"""Drop-in mirror of `youtokentome.BPE` / `youtokentome.OutputType` (reference: youtokentome/youtokentome.py:6-99) and
of the Cython class it wraps (`_youtokentome_cython.BPE`, youtokentome/cpp/yttm.pyx:51-182), on top of the MI355X C ABI
(include/yttm_mi355x.h).  Same names, argument meaning, return shapes and error behaviour (ValueError / TypeError)."""
import ctypes as C
from collections.abc import Collection
from enum import Enum
from typing import List, Optional, Union

import numpy as np

from . import _lib

try:  # csrc/pyapi.c, built next to the HIP library; without it the list API marshals in Python (slower, same results)
    from . import _yttm_pyapi as _pyapi
except ImportError:  # pragma: no cover
    _pyapi = None


class OutputType(Enum):  # youtokentome.py:6-8
    ID = 1
    SUBWORD = 2


def _err():
    return C.create_string_buffer(_lib.ERRLEN)


def _pack(sentences):
    """list[str] -> (utf-8 blob, uint64 offsets[n+1])  (the `.encode()` per sentence of yttm.pyx:103)"""
    offs = np.zeros(len(sentences) + 1, dtype=np.uint64)
    if not sentences:
        return b"", offs
    blob = "".join(sentences).encode()
    nchar = np.fromiter(map(len, sentences), dtype=np.int64, count=len(sentences))
    if int(nchar.sum()) == len(blob):  # one byte per char everywhere: the byte offsets are the char offsets, no per-sentence encode
        np.cumsum(nchar, out=offs[1:])
        return blob, offs
    enc = [s.encode() for s in sentences]
    np.cumsum(np.fromiter(map(len, enc), dtype=np.int64, count=len(enc)), out=offs[1:])
    return b"".join(enc), offs


def _take(ptr, n, ctype, dtype):
    L = _lib.load()
    n = int(n)
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(max(n, 1),))[:n].astype(dtype, copy=True)
    L.yttm_free(C.cast(ptr, C.c_void_p))
    return arr


class _Core:
    """`_youtokentome_cython.BPE` (yttm.pyx:51-182)."""

    def __init__(self, model_path, n_threads=-1, device=0):
        L = _lib.load()
        h = C.c_void_p()
        err = _err()
        rc = L.yttm_encoder_create(model_path.encode(), n_threads, device, C.byref(h), err, _lib.ERRLEN)
        if rc != 0:
            raise ValueError(err.value.decode())  # yttm.pyx:61-62
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.load().yttm_encoder_destroy(h)
            self._h = None

    @staticmethod
    def train(data, model, vocab_size, coverage=1.0, n_threads=-1, pad_id=0, unk_id=1, bos_id=2, eos_id=3):
        err = _err()
        rc = _lib.load().yttm_train_bpe(data.encode(), model.encode(), vocab_size, coverage, n_threads, pad_id, unk_id,
                                        bos_id, eos_id, err, _lib.ERRLEN)
        if rc != 0:
            raise ValueError(err.value.decode())  # yttm.pyx:84-85

    # ---- word cache of the batch encoder (include/yttm_mi355x.h): 0 off, 1 whenever possible, 2 (default) from min_bytes up
    def set_cache(self, mode, min_bytes=8 << 20):
        _lib.load().yttm_encoder_set_cache(self._h, int(mode), int(min_bytes))

    def cache_words(self):
        return int(_lib.load().yttm_encode_cache_words(self._h))

    # ---- packed fast path: bytes + offsets -> numpy ids + offsets
    def encode_packed(self, blob: bytes, offsets, bos=False, eos=False, reverse=False, dropout_prob=0.0):
        L = _lib.load()
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        ids, off = _lib.i32p(), _lib.u64p()
        err = _err()
        rc = L.yttm_encode_as_ids(self._h, blob, offsets.ctypes.data_as(_lib.u64p), n, int(bos), int(eos), int(reverse),
                                  float(dropout_prob), C.byref(ids), C.byref(off), err, _lib.ERRLEN)
        if rc != 0:
            raise ValueError(err.value.decode())
        off_a = _take(off, n + 1, C.c_uint64, np.uint64)
        ids_a = _take(ids, int(off_a[-1]), C.c_int32, np.int32)
        return ids_a, off_a

    def encode(self, sentences, output_type, bos, eos, reverse, dropout_prob):
        if dropout_prob < 0 or dropout_prob > 1:  # yttm.pyx:92-93
            raise ValueError("dropout_prob value must be in the range [0, 1]. Current value of dropout_prob = " + str(dropout_prob))
        single = isinstance(sentences, str)
        if single:
            batch = [sentences]
        else:
            assert isinstance(sentences, list) or isinstance(sentences, tuple)
            batch = list(sentences)
        if output_type == "id" and _pyapi is not None:
            # list[str] -> blob + offsets -> yttm_encode_as_ids -> list[list[int]] in C (csrc/pyapi.c: what yttm.pyx:87-109 does in Cython)
            L = _lib.load()
            out = _pyapi.encode_ids(C.cast(L.yttm_encode_as_ids, C.c_void_p).value, C.cast(L.yttm_free, C.c_void_p).value, self._h.value,
                                    batch, bool(bos), bool(eos), bool(reverse), float(dropout_prob))
            return out[0] if single else out
        blob, offs = _pack(batch)
        if output_type == "id":
            ids, off = self.encode_packed(blob, offs, bos, eos, reverse, dropout_prob)
            flat, o = ids.tolist(), off.tolist()  # (slicing a list of ints is several times faster than numpy slice + tolist per sentence)
            out = [flat[o[i]:o[i + 1]] for i in range(len(batch))]
        elif output_type == "subword":
            L = _lib.load()
            blob_p, poff, soff = C.c_void_p(), _lib.u64p(), _lib.u64p()
            npieces = C.c_uint64()
            err = _err()
            rc = L.yttm_encode_as_subwords(self._h, blob, offs.ctypes.data_as(_lib.u64p), len(batch), int(bos), int(eos),
                                           int(reverse), float(dropout_prob), C.byref(blob_p), C.byref(poff),
                                           C.byref(npieces), C.byref(soff), err, _lib.ERRLEN)
            if rc != 0:
                raise ValueError(err.value.decode())
            po = _take(poff, npieces.value + 1, C.c_uint64, np.uint64)
            so = _take(soff, len(batch) + 1, C.c_uint64, np.uint64)
            raw = C.string_at(blob_p, int(po[-1]))
            L.yttm_free(blob_p)
            pieces = [raw[int(po[i]):int(po[i + 1])].decode() for i in range(npieces.value)]
            out = [pieces[int(so[i]):int(so[i + 1])] for i in range(len(batch))]
        else:
            raise ValueError('output_type must be equal to "id" or "subword"')  # yttm.pyx:124
        return out[0] if single else out

    def subword_to_id(self, subword):
        return _lib.load().yttm_subword_to_id(self._h, subword.encode())

    def id_to_subword(self, id):
        L = _lib.load()
        p = C.c_void_p()
        err = _err()
        rc = L.yttm_id_to_subword(self._h, int(id), C.byref(p), err, _lib.ERRLEN)
        if rc != 0:
            raise ValueError(err.value.decode())
        s = C.string_at(p).decode()
        L.yttm_free(p)
        return s

    def decode(self, ids, ignore_ids):
        if not isinstance(ids, list):  # yttm.pyx:138-141
            raise TypeError("{} is not a list instance".format(type(ids)))
        if not isinstance(ignore_ids, Collection) and ignore_ids is not None:
            raise TypeError("{} is not a Collection instance".format(type(ignore_ids)))
        if len(ids) > 0 and isinstance(ids[0], int):
            ids = [ids]
        if ignore_ids is None:
            ignore_ids = set()
        L = _lib.load()
        flat = np.ascontiguousarray([t for s in ids for t in s], dtype=np.int32)
        offs = np.zeros(len(ids) + 1, dtype=np.uint64)
        if ids:
            np.cumsum([len(s) for s in ids], out=offs[1:])
        ign = np.ascontiguousarray(sorted(set(int(i) for i in ignore_ids)), dtype=np.int32)
        blob_p, ooff = C.c_void_p(), _lib.u64p()
        err = _err()
        rc = L.yttm_decode(self._h, flat.ctypes.data_as(_lib.i32p), offs.ctypes.data_as(_lib.u64p), len(ids),
                           ign.ctypes.data_as(_lib.i32p), len(ign), C.byref(blob_p), C.byref(ooff), err, _lib.ERRLEN)
        if rc != 0:
            raise ValueError(err.value.decode())
        oo = _take(ooff, len(ids) + 1, C.c_uint64, np.uint64)
        raw = C.string_at(blob_p, int(oo[-1]))
        L.yttm_free(blob_p)
        return [raw[int(oo[i]):int(oo[i + 1])].decode() for i in range(len(ids))]

    def vocab_size(self):
        return _lib.load().yttm_vocab_size(self._h)

    # ---- the command line's streaming loops (yttm.pyx:167-181): stdin -> stdout inside the library
    def encode_cli(self, output_type, stream, bos, eos, reverse, dropout_prob, in_fd=0, out_fd=1):
        import sys
        sys.stdout.flush()
        err = _err()
        rc = _lib.load().yttm_encode_cli(self._h, output_type.encode(), int(stream), int(bos), int(eos), int(reverse),
                                         float(dropout_prob), in_fd, out_fd, err, _lib.ERRLEN)
        if rc != 0:
            raise ValueError(err.value.decode())

    def decode_cli(self, ignore_ids, in_fd=0, out_fd=1):
        import sys
        sys.stdout.flush()
        ign = np.ascontiguousarray(sorted(set(int(i) for i in (ignore_ids or ()))), dtype=np.int32)
        err = _err()
        rc = _lib.load().yttm_decode_cli(self._h, ign.ctypes.data_as(_lib.i32p), len(ign), in_fd, out_fd, err, _lib.ERRLEN)
        if rc != 0:
            raise ValueError(err.value.decode())

    def vocab_cli(self, verbose, out_fd=1):
        import sys
        sys.stdout.flush()
        err = _err()
        rc = _lib.load().yttm_vocab_cli(self._h, int(verbose), out_fd, err, _lib.ERRLEN)
        if rc != 0:
            raise ValueError(err.value.decode())

    def vocab(self):
        L = _lib.load()
        blob_p, off = C.c_void_p(), _lib.u64p()
        n = C.c_uint64()
        L.yttm_vocabulary(self._h, C.byref(blob_p), C.byref(off), C.byref(n))
        oo = _take(off, n.value + 1, C.c_uint64, np.uint64)
        raw = C.string_at(blob_p, int(oo[-1]))
        L.yttm_free(blob_p)
        return [raw[int(oo[i]):int(oo[i + 1])].decode() for i in range(n.value)]


class BPE:
    """youtokentome.BPE (youtokentome.py:11-99)."""

    def __init__(self, model: str, n_threads: int = -1):
        self.model = model
        self.n_threads = n_threads
        self.bpe_cython = _Core(model_path=model, n_threads=n_threads)

    @staticmethod
    def train(data: str, model: str, vocab_size: int, coverage: float = 1.0, n_threads: int = -1, pad_id: int = 0,
              unk_id: int = 1, bos_id: int = 2, eos_id: int = 3) -> "BPE":
        _Core.train(data=data, model=model, vocab_size=vocab_size, n_threads=n_threads, coverage=coverage, pad_id=pad_id,
                    unk_id=unk_id, bos_id=bos_id, eos_id=eos_id)
        return BPE(model=model, n_threads=n_threads)

    def encode(self, sentences: List[str], output_type: OutputType = OutputType.ID, bos: bool = False, eos: bool = False,
               reverse: bool = False, dropout_prob: float = 0) -> Union[List[List[int]], List[List[str]]]:
        if not isinstance(output_type, OutputType):
            raise TypeError("parameter output_type must be youtokentome.OutputType, not %s}" % str(type(output_type)))
        output_type_str = "id" if output_type == OutputType.ID else "subword"
        return self.bpe_cython.encode(sentences=sentences, output_type=output_type_str, bos=bos, eos=eos, reverse=reverse,
                                      dropout_prob=dropout_prob)

    def vocab_size(self) -> int:
        return self.bpe_cython.vocab_size()

    def vocab(self) -> List[str]:
        return self.bpe_cython.vocab()

    def subword_to_id(self, subword: str) -> int:
        return self.bpe_cython.subword_to_id(subword)

    def id_to_subword(self, id: int) -> str:
        return self.bpe_cython.id_to_subword(id)

    def decode(self, ids: Union[List[int], List[List[int]]], ignore_ids: Optional[Collection] = None) -> List[str]:
        return self.bpe_cython.decode(ids, ignore_ids)

    def __getstate__(self):
        return {"model": self.model, "n_threads": self.n_threads}

    def __setstate__(self, dict):
        self.model = dict["model"]
        self.n_threads = dict["n_threads"]
        self.bpe_cython = _Core(model_path=self.model, n_threads=self.n_threads)
