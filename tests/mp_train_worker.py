"""Worker of the world_size-2 CPU test: one process per rank, torch.distributed (gloo) as the transport behind the
library's host-callback communicator, product sources running under the HIP emulator.  Usage:
    python mp_train_worker.py <rank> <world> <port> <corpus> <model_out> <vocab> <coverage>"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def make_callbacks(L, dist, torch, rank, world):
    def allreduce(user, buf, n):
        arr = np.ctypeslib.as_array(buf, shape=(n,))
        t = torch.from_numpy(arr.view(np.int64))
        dist.all_reduce(t)  # uint64 sum == int64 sum modulo 2^64
        return 0

    def allgather(user, send, nbytes, recv, cap, out_bytes):
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([nbytes], dtype=torch.int64))
        sizes = [int(s.item()) for s in sizes]
        mx = max(max(sizes), 1)
        mine = torch.zeros(mx, dtype=torch.uint8)
        if nbytes:
            mine[:nbytes] = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,)).copy())
        parts = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(parts, mine)
        blob = b"".join(parts[r][: sizes[r]].numpy().tobytes() for r in range(world) if r != rank)
        out_bytes[0] = len(blob)
        if len(blob) <= cap:
            C.memmove(recv, blob, len(blob))
        return 0

    return L.ALLREDUCE_FN(allreduce), L.ALLGATHER_FN(allgather)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    corpus, model_out, vocab, coverage = sys.argv[4], sys.argv[5], int(sys.argv[6]), float(sys.argv[7])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    if os.environ.get("YTTM_TEST_FREE_BYTES_RANK%d" % rank):  # (tests: this rank alone is short of device memory)
        os.environ["YTTM_TEST_FREE_BYTES"] = os.environ["YTTM_TEST_FREE_BYTES_RANK%d" % rank]
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from youtokentome_amd import _lib
    L = _lib.load()
    text = open(corpus, "rb").read()

    def split(i):  # the reference's per-thread split: advance to the next ASCII space (bpe.cpp:864-873)
        if i == 0:
            return 0
        c = len(text) * i // world
        while c < len(text) and text[c] not in b" \t\n\v\f\r":
            c += 1
        return c
    shard = text[split(rank):split(rank + 1)]
    ar, ag = make_callbacks(L, dist, torch, rank, world)
    comm = C.c_void_p()
    assert L.yttm_comm_callback_create(rank, world, ar, ag, None, C.byref(comm)) == 0
    err = C.create_string_buffer(2048)
    rep = C.create_string_buffer(16384)
    rc = L.yttm_train_bpe_from_memory_comm(shard, len(shard), model_out.encode() if rank == 0 else b"", vocab, coverage, 0, 1, 2, 3, 0,
                                           comm, rep, 16384, err, 2048)
    L.yttm_comm_destroy(comm)
    dist.barrier()
    dist.destroy_process_group()
    if rc != 0:
        print("ERR", err.value.decode())
        sys.exit(3)
    expect = os.environ.get("YTTM_TEST_EXPECT", "")
    if rank == 0 and os.environ.get("YTTM_TEST_EXPECT_RANK0"):  # (checks that only hold on a rank that is sure to have words)
        expect = ",".join(filter(None, [expect, os.environ["YTTM_TEST_EXPECT_RANK0"]]))
    for want in filter(None, expect.split(",")):  # e.g. "word_rounds>0,word_fused_rounds==0": checks on this rank's report
        import json
        import re
        key, op, val = re.match(r"(\w+)(>|==)(\d+)$", want).groups()
        got = json.loads(rep.value.decode())[key]
        if not (got > int(val) if op == ">" else got == int(val)):
            print("ERR report", want, "got", got)
            sys.exit(4)
    print("OK", rank)


if __name__ == "__main__":
    main()
