#!/usr/bin/env python3
"""bench.py -- BPE.train MB/s (+ encode sentences/s) on MI355X, vocab 32000  (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" = one full BPE.train() pass (char histogram -> word dedup -> pair count -> merge loop -> model file) over
one synthetic corpus that is ALREADY RESIDENT IN HBM when the timed region starts.  N=1 workload = BASELINE.json
configs[1]: 1 GB random 'abcd ' corpus, vocab_size=32000.  N>1: one process per GPU (torchrun), every rank holds its
own 1 GB shard (weak scaling), pair-count deltas are exchanged over RCCL each round.  Rank 0 prints ONE JSON line.
The same line carries: `roofline` (dominant kernel, HIP-event timed inside the timed region), `kernels` (per kernel
family), `encode` (sentences/s of the batch-encode kernel on 128-char sentences with the freshly trained model) and
`cpu_baseline` (the UNMODIFIED reference, oracle/_ref/yttm_ref_prod, n_threads=8, timed on this box's host cores on a
bounded sample)."""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size-mb", type=int, default=1000, help="corpus MB per GPU (1000 = BASELINE configs[1])")
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--corpus", default="abcd", choices=["abcd", "zipf"])
    ap.add_argument("--encode-sentences", type=int, default=10_000_000)
    ap.add_argument("--no-encode", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mb", type=int, default=100)
    args = ap.parse_args()

    import numpy as np
    import torch

    import gen
    from youtokentome_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        log(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE")
    n_gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(backend="nccl", device_id=dev)
    L = _lib.load()
    rc, info = _lib.device_info(local_rank)
    if rank == 0:
        log("device:", info)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- synthetic corpus shard, resident in HBM --------------------------------------------------------------------
    nbytes = args.size_mb * 1_000_000
    t0 = time.time()
    if args.corpus == "abcd":
        host = gen.abcd_corpus(nbytes, seed=19 + rank, survey_stream=True)  # rank 0, 1000 MB: SURVEY.md's C2 file byte for byte (md5 63857720...)
    else:
        host = gen.zipf_corpus(nbytes, seed=7 + rank, vocab=400000)
    corpus = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
    n_local = corpus.numel()
    if rank == 0:
        log(f"corpus: {args.corpus} {n_local/1e6:.1f} MB per GPU generated+uploaded in {time.time()-t0:.1f}s")
    comm_handle = None
    if world > 1:
        comm_handle = _init_rccl(L, dist, torch, dev, rank, world, local_rank)

    tmpdir = tempfile.mkdtemp(prefix="yttm_bench_")
    # one model file per job: rank 0 writes it (single node), every rank's encoder loads it afterwards
    model_path = os.path.join(tempfile.gettempdir(), "yttm_bench_%s.model" % os.environ.get("MASTER_PORT", str(os.getpid())))
    err = C.create_string_buffer(_lib.ERRLEN)
    rep = C.create_string_buffer(8192)

    def train_step(profile):
        if comm_handle is not None:
            rc = L.yttm_train_bpe_from_device_comm(C.c_void_p(corpus.data_ptr()), n_local, model_path.encode(), args.vocab, 1.0,
                                                   0, 1, 2, 3, local_rank, int(profile), comm_handle, rep, 8192, err, _lib.ERRLEN)
        else:
            rc = L.yttm_train_bpe_from_device(C.c_void_p(corpus.data_ptr()), n_local, model_path.encode(), args.vocab, 1.0,
                                              0, 1, 2, 3, local_rank, int(profile), rep, 8192, err, _lib.ERRLEN)
        if rc != 0:
            raise RuntimeError("train failed: " + err.value.decode())
        return json.loads(rep.value.decode())

    devnull = os.open(os.devnull, os.O_WRONLY)
    saved_err = os.dup(2)

    def quiet(on):  # the trainer prints progress to stderr like the reference
        os.dup2(devnull if on else saved_err, 2)

    quiet(not os.environ.get("YTTM_TRACE"))
    try:
        for _ in range(args.warmup):
            train_step(False)
        barrier()
        t0 = time.perf_counter()
        reports = []
        for _ in range(args.steps):
            reports.append(train_step(True))
        barrier()
        dt = time.perf_counter() - t0
    finally:
        quiet(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_bytes = n_local * n_gpus
    value = args.steps * total_bytes / dt / 1e6
    pins = {}  # md5 of rank 0's corpus shard (SURVEY.md section 8d pins the generated files) and of the model file it produced
    if rank == 0:
        try:
            import hashlib
            pins["corpus_md5"] = hashlib.md5(host).hexdigest()
            pins["model_md5"] = hashlib.md5(open(model_path, "rb").read()).hexdigest()
        except Exception as e:  # never fail the bench line over a checksum
            pins["pins_error"] = str(e)
    r = reports[-1]

    # ---- per-kernel roofline from the HIP-event times collected inside the timed steps ----------------------------------
    kern = {}
    for name, k in r["kernels"].items():
        if k["launches"] and k["ms"] > 0:
            kern[name] = {"ms_total": round(k["ms"], 3), "launches": k["launches"],
                          "avg_ms": round(k["ms"] / k["launches"], 4),
                          "algorithmic_GB": round(k["bytes"] / 1e9, 4),
                          "GBps": round(k["bytes"] / 1e9 / (k["ms"] / 1e3), 1)}
    # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2-corrected + WRITE_SIZE, separate
    # --pmc runs of this same command; tools/pmc_summary.py) -- only quoted for the exact workload they were taken on
    traffic = {}
    pmc_file = os.path.join(ROOT, "profiles", "r1_1gb_final_pmc_hbm.json")  # written by tools/profile_round.sh
    if os.path.exists(pmc_file) and args.size_mb == 1000 and args.corpus == "abcd" and args.vocab == 32000 and n_gpus == 1:
        pm = json.load(open(pmc_file))
        # kernel families as the trainer times them; a family's traffic per launch = bytes of all its kernels / its launches
        def family(k):  # k_tiles<SLOT, WPB, MERGE, LDSR>: MERGE=false is K3 (pair count), true is K4 (merge apply)
            if k.startswith("k_tiles<"):
                return "merge_apply" if k[len("k_tiles<"):].split(", ")[2] == "true" else "pair_count"
            for name, prefixes in (("char_hist", ("k_scan_bytes<0>",)), ("segments", ("k_scan_bytes<1>",)), ("dedup", ("k2b_insert_words",)),
                                   ("merge_apply", ("k_filter<", "k_giant<true")), ("pair_count", ("k_giant<false",)),
                                   ("cand_scan", ("k_hot_scan", "k_cand_scan"))):
                if k.startswith(prefixes):
                    return name
            return None
        for name in ("char_hist", "segments", "dedup", "pair_count", "merge_apply", "cand_scan"):
            ks = [k for k in pm if family(k) == name]
            if ks and name in kern:
                total = sum(pm[k]["traffic_bytes_per_launch"] * pm[k]["launches"] for k in ks)
                traffic[name] = round(total / max(1, kern[name]["launches"]))  # per launch of the family as the trainer counts them
    dom = max(kern, key=lambda n: kern[n]["ms_total"]) if kern else None
    roofline = None
    if dom:
        roofline = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(kern[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic.get(dom),
                    "traffic_unit": "bytes per launch (rocprofv3 PMC, profiles/r1_1gb_final_pmc_hbm.json)",
                    "algorithmic_bytes_per_launch": round(kern[dom]["algorithmic_GB"] * 1e9 / kern[dom]["launches"]),
                    "avg_launch_ms": kern[dom]["avg_ms"], "launches": kern[dom]["launches"]}
    roofline_pc = None
    if "pair_count" in kern:
        roofline_pc = {"kernel": "pair_count (K3)", "bound": "hbm", "achieved": kern["pair_count"]["GBps"], "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(kern["pair_count"]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic.get("pair_count"),
                       "algorithmic_bytes_per_launch": round(kern["pair_count"]["algorithmic_GB"] * 1e9)}

    out = {
        "metric": "bpe_train_throughput", "value": round(value, 2), "unit": "MB/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"{args.size_mb} MB/GPU random '{'abcd ' if args.corpus == 'abcd' else 'zipf ascii'}' corpus, "
                               f"vocab_size={args.vocab} (BASELINE.json configs[1])",
                   "corpus_bytes_per_gpu": n_local, "vocab_size": args.vocab, "unique_words": r["n_unique"],
                   "dedup_tokens": r["n_tokens"], "merge_rounds": r["rounds"], "rules": r["rules"],
                   "input": "resident in HBM before the timed region", **pins},
        "roofline": roofline, "roofline_pair_count": roofline_pc, "kernels": kern,
        "phases_s": {"frontend": r["seconds_frontend"], "merge_loop": r["seconds_merge"], "dump": r["seconds_io"]},
    }

    # ---- encode: sentences/s on 128-char sentences with the model just trained (BASELINE configs[3]) ---------------------
    if not args.no_encode:
        out["encode"] = _bench_encode(L, _lib, torch, np, gen, dev, local_rank, model_path, args, rank, world, dist, barrier)

    # ---- CPU baseline: the unmodified reference on this box's host cores (rank 0, N=1 only, bounded sample) --------------
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = _cpu_baseline(host, args, tmpdir, out.get("encode"))
    if "encode" in out:
        out["encode"].pop("_host_sample", None)
        out["encode"].pop("_sample_ids", None)
        out["encode"].pop("_model_path", None)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _bench_encode(L, _lib, torch, np, gen, dev, local_rank, model_path, args, rank, world, dist, barrier):
    n_sent = args.encode_sentences
    line = 128
    host = gen.abcd_corpus(n_sent * (line + 1), seed=123 + rank, line=line, survey_stream=True)  # SURVEY.md's C4 stream
    n_sent = len(host) // (line + 1)
    d_bytes = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
    d_off = (torch.arange(n_sent + 1, dtype=torch.int64, device=dev) * (line + 1))
    err = C.create_string_buffer(_lib.ERRLEN)
    h = C.c_void_p()
    # every rank loads the model written by rank 0's trainer (identical on all ranks by construction)
    if L.yttm_encoder_create(model_path.encode(), 1, local_rank, C.byref(h), err, _lib.ERRLEN) != 0:
        raise RuntimeError(err.value.decode())
    n_ids = C.c_uint64()
    kms = C.c_double()

    def step():
        rc = L.yttm_encode_device(h, C.c_void_p(d_bytes.data_ptr()), C.c_void_p(d_off.data_ptr()), n_sent, d_bytes.numel(),
                                  line + 1, 0, 0, 0, 0.0, C.byref(n_ids), C.byref(kms), err, _lib.ERRLEN)
        if rc != 0:
            raise RuntimeError(err.value.decode())
    step()
    barrier()
    t0 = time.perf_counter()
    k_ms = []
    for _ in range(max(1, args.steps)):
        step()
        k_ms.append(kms.value)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    steps = max(1, args.steps)
    sps = steps * n_sent * world / dt
    alg_bytes = d_bytes.numel() + 8 * (n_sent + 1) * 2 + 4 * n_ids.value  # B_in + 16(S+1) + 4 K_out
    kavg = sum(k_ms) / len(k_ms)
    res = {"metric": "encode_sentences_per_s", "value": round(sps, 1), "unit": "sentences/s", "sentences_per_gpu": n_sent,
           "sentence_chars": line, "ids_per_sentence": round(n_ids.value / n_sent, 3), "ms_per_step": round(dt / steps * 1e3, 2),
           "kernel_ms": round(kavg, 3),
           "roofline": {"kernel": "k5_encode", "bound": "hbm", "achieved": round(alg_bytes / 1e9 / (kavg / 1e3), 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(alg_bytes / 1e9 / (kavg / 1e3) / HBM_PEAK_GBS, 4), "traffic": None}}
    # FNV-1a-64 of (len, ids...) per sentence over a bounded sample, for the parity line next to the CPU baseline
    res["_host_sample"] = host[: 1_000_000 * (line + 1)]
    res["_model_path"] = model_path
    m = min(n_sent, 1_000_000)
    ids = np.zeros(n_ids.value, dtype=np.int32)
    off = np.zeros(n_sent + 1, dtype=np.uint64)
    L.yttm_encode_fetch(h, ids.ctypes.data_as(_lib.i32p), off.ctypes.data_as(_lib.u64p), n_sent, err, _lib.ERRLEN)
    res["_sample_ids"] = (ids[: int(off[m])], off[: m + 1])
    L.yttm_encoder_destroy(h)
    return res


def _fnv(ids, off):
    import numpy as np
    h = 1469598103934665603
    mask = (1 << 64) - 1
    ids = ids.astype(np.uint32)
    for i in range(len(off) - 1):
        a, b = int(off[i]), int(off[i + 1])
        for v in [b - a] + ids[a:b].tolist():
            for k in range(4):
                h ^= (v >> (8 * k)) & 0xff
                h = (h * 1099511628211) & mask
    return "%016x" % h


def _cpu_baseline(host, args, tmpdir, enc):
    import subprocess
    ref = os.path.join(ROOT, "oracle", "_ref", "yttm_ref_prod")
    if not os.path.exists(ref):
        return {"value": None, "unit": "MB/s", "cores": 0, "kind": "reference", "sample": "oracle/_ref/yttm_ref_prod not built"}
    sample = host[: args.cpu_sample_mb * 1_000_000]
    sample = sample[: sample.rfind(b"\n") + 1]
    path = os.path.join(tmpdir, "cpu_sample.txt")
    with open(path, "wb") as f:
        f.write(sample)
    model = os.path.join(tmpdir, "cpu_sample.model")
    r = subprocess.run([ref, "train", path, model, str(args.vocab), "1.0", "8", "0", "1", "2", "3"], capture_output=True, text=True)
    res = {"value": None, "unit": "MB/s", "cores": 8, "kind": "reference",
           "sample": f"unmodified reference (oracle/_ref/yttm_ref_prod = bpe.cpp as shipped, -O3), n_threads=8, train on the first "
                     f"{len(sample)/1e6:.0f} MB of the same corpus, vocab {args.vocab}; C++ boundary (train_bpe)"}
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1])
        res["value"] = round(len(sample) / 1e6 / j["train_seconds"], 2)
        res["train_seconds"] = j["train_seconds"]
    except Exception as e:  # noqa: BLE001
        res["error"] = str(e)
    if enc is not None and "_host_sample" in enc:
        lines = os.path.join(tmpdir, "enc_sample.txt")
        with open(lines, "wb") as f:
            f.write(enc["_host_sample"])
        r = subprocess.run([ref, "encode_bench", enc["_model_path"], lines, "8", "0.0", "1000000"], capture_output=True, text=True)
        try:
            j = json.loads(r.stdout.strip().splitlines()[-1])
            ids, off = enc["_sample_ids"]
            res["encode"] = {"value": round(j["sentences"] / j["encode_seconds"], 1), "unit": "sentences/s", "cores": 8,
                             "sample": f"{j['sentences']} sentences of 128 chars, encode_as_ids, n_threads=8",
                             "ids_match_gpu": _fnv(ids[: int(off[j['sentences']])], off[: j['sentences'] + 1]) == j["fnv1a64"]}
        except Exception as e:  # noqa: BLE001
            res["encode"] = {"error": str(e)}
    if enc is not None:
        enc.pop("_host_sample", None)
        enc.pop("_sample_ids", None)
        enc.pop("_model_path", None)
    return res


def _init_rccl(L, dist, torch, dev, rank, world, local_rank):
    """RCCL communicator for the library: rank 0 creates the unique id, torch.distributed broadcasts it."""
    idbuf = (C.c_uint8 * 128)()
    if rank == 0:
        if L.yttm_comm_rccl_unique_id(idbuf) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
    t = torch.tensor(list(idbuf), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)
    idbuf = (C.c_uint8 * 128)(*t.cpu().tolist())
    h = C.c_void_p()
    if L.yttm_comm_rccl_create(idbuf, rank, world, local_rank, C.byref(h)) != 0:
        raise RuntimeError("ncclCommInitRank failed")
    return h


if __name__ == "__main__":
    main()
