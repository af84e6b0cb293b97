"""Worker of the GPU ordering test (tests/test_gpu_parity.py::test_zz_fused_tail_ordering): one training per process, so that every run
has its own YTTM_DBG_CAND trace file and its own environment hooks.  Usage:
    python gpu_train_worker.py <corpus> <model_out> <vocab> <comm: 0 | 1 (an RCCL communicator of one rank: the multi-GPU round protocol)>"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    corpus, model, vocab, use_comm = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    from youtokentome_amd import _lib
    L = _lib.load()
    err, rep = C.create_string_buffer(2048), C.create_string_buffer(16384)
    comm = C.c_void_p()
    if use_comm:
        idbuf = (C.c_uint8 * 128)()
        assert L.yttm_comm_rccl_unique_id(idbuf) == 0
        assert L.yttm_comm_rccl_create(idbuf, 0, 1, 0, C.byref(comm)) == 0
    rc = L.yttm_train_bpe_comm(corpus.encode(), model.encode(), vocab, 1.0, 8, 0, 1, 2, 3, 0, 0, comm if use_comm else None, rep, 16384, err, 2048)
    if use_comm:
        L.yttm_comm_destroy(comm)
    if rc != 0:
        print("ERR", err.value.decode())
        sys.exit(3)
    print("REPORT " + rep.value.decode(), flush=True)  # (RCCL prints a banner of its own to stdout)


if __name__ == "__main__":
    main()
