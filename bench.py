#!/usr/bin/env python3
"""bench.py -- BPE.train MB/s (+ encode sentences/s) on MI355X, vocab 32000  (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" = one full BPE.train() -- SURVEY.md 8d's metric: file bytes / wall from the call to the model file closed.  The corpus FILE
sits in the page cache (like the CPU baseline's input); a step is yttm_train_bpe_comm(path -> model): open, pread into pinned chunks,
H2D, char histogram -> word dedup -> pair count -> merge loop -> model file.  That is `value`.  The same steps with the corpus ALREADY
RESIDENT IN HBM are timed right after and reported beside it (`value_hbm_resident`; the kernel table and the roofline come from the
timed file steps -- the kernels are the same).  Workload = BASELINE.json configs[1]: the 1 GB random 'abcd ' corpus of SURVEY.md
Appendix C (seed 19, md5 pinned), vocab_size=32000.  N>1: one process per GPU (torchrun); every rank reads ITS byte range of the SAME
pinned file, cut at whitespace (strong scaling -- the only way the N-GPU model can be compared with the reference's), pair-count
deltas travel over RCCL each round.  `--scaling weak` gives every rank its own 1 GB instead (HBM-resident only; no pin exists for those
corpora).  Rank 0 prints ONE JSON line.

Parity is part of the line: the md5 of every model the GPU writes (configs[1], the Zipf corpus of configs[2]) and the
FNV of all 10 M encoded sentences (configs[3]) are compared with tests/golden/full_size_pins.json -- outputs of the
UNMODIFIED reference (tests/golden/make_full_pins.py) -- and a mismatch makes the process exit non-zero.

Also on the line: `roofline` (dominant kernel; `achieved` = the contract's algorithmic bytes 8*T_touched + 8*W_touched, counted at word
granularity by a separate untimed measurement pass, over the kernel's time by HIP events -- the device-clock figure of the timed steps beside it),
`roofline_pair_count` (K3, the kernel north_star names), `kernels`, `e2e` (host -> host encode, the Python list API), `encode` / `encode_dropout` (configs[3] / [4]), `extra.zipf` (configs[2]) and
`cpu_baseline` (the unmodified reference, n_threads=8, pinned to 8 cores, on the SAME full inputs)."""
import argparse
import ctypes as C
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md) -- what `frac` is of
HBM_COPY_GBS = 6290.0  # the measured float4-copy ceiling of the same guide ("report fraction of both") -- `frac_of_measured_copy`
PINS = os.path.join(ROOT, "tests", "golden", "full_size_pins.json")


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def both_fracs(r):
    """A roofline object's `frac` (of the 8 TB/s spec) gets `frac_of_measured_copy` (of the 6.29 TB/s a copy kernel reaches) beside it."""
    if isinstance(r, dict) and isinstance(r.get("achieved"), (int, float)):
        r["frac_of_measured_copy"] = round(r["achieved"] / HBM_COPY_GBS, 4)
        r["peak_measured_copy"] = HBM_COPY_GBS
    return r


def profile_for(tag, suffix):
    """The newest profiles/rN_<tag>_<suffix> (N = round number) -- a file is only QUOTED when its `_meta.source_sha16` (or its sibling
    PMC summary's) names this build's sources, see _static_traffic."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_%s_%s" % (tag, suffix))):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return best[1] if best else None


def sources_sha():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_summary import source_sha16
    return source_sha16(ROOT)


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def pin_name(corpus, size_mb):
    return {("abcd", 1000): "c2_1gb", ("abcd", 100): "c2_100mb", ("zipf", 1000): "c3_1gb", ("zipf", 100): "c3_100mb",
            ("cjk", 1000): "c6_cjk_1gb", ("cjk", 100): "c6_cjk_100mb", ("zipf4m", 1000): "c7_zipf4m_1gb"}.get((corpus, size_mb))


def split_points(host, world):
    """Byte ranges per rank: size*r/W advanced to the next ASCII white space (host_trainer.cpp train_bpe; bpe.cpp:864-873)."""
    n = len(host)
    cuts = [0]
    for r in range(1, world):
        c = n * r // world
        while c < n and host[c] not in b" \t\n\v\f\r":
            c += 1
        cuts.append(c)
    cuts.append(n)
    return cuts


class Quiet:
    """The trainer prints progress to stderr like the reference; keep the bench log readable."""

    def __init__(self):
        self.devnull = os.open(os.devnull, os.O_WRONLY)
        self.saved = os.dup(2)
        self.on = not os.environ.get("YTTM_TRACE")

    def __enter__(self):
        if self.on:
            os.dup2(self.devnull, 2)

    def __exit__(self, *a):
        os.dup2(self.saved, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)  # (five: the mean of two file -> model steps moved by 4 % with one slow upload among them)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size-mb", type=int, default=1000, help="corpus MB (1000 = BASELINE configs[1])")
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--corpus", default="abcd", choices=["abcd", "zipf", "cjk"])  # (zipf / cjk as the main workload: tools/profile_round.sh variants)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N>1: shard ONE pinned corpus (strong) or one corpus per GPU (weak)")
    ap.add_argument("--encode-sentences", type=int, default=10_000_000)
    ap.add_argument("--no-encode", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the Zipf (configs[2]) block")
    ap.add_argument("--no-extra2", action="store_true", help="skip the large-alphabet (CJK-shaped) and enwik-like (4e6-word lexicon) training blocks")
    ap.add_argument("--watchdog-seconds", type=int, default=-1,
                    help="N > 1: every rank prints an error line and exits non-zero if the run has not finished by then (default: 1500 for N > 1, off for one GPU). "
                         "The RCCL exchange has never run with more than one rank on hardware; a hang must end as a verdict, not as the driver's time-out")
    ap.add_argument("--no-big", action="store_true", help="skip the beyond-2^32 block: the 8.8 GB corpus with a word seen 4.4e9 times (file -> model, pinned)")
    ap.add_argument("--big-zipf", action="store_true", help="also the 8 GB Zipf corpus (its generation alone takes two minutes: profiles/ holds a run)")
    ap.add_argument("--no-touched-pass", action="store_true", help="skip the untimed K4 measurement pass (profiling runs: one training per process)")
    ap.add_argument("--cpu-sample-mb", type=int, default=0, help="0 = the full corpus")
    ap.add_argument("--cpu-runs", type=int, default=3, help="runs of the reference's train_bpe per corpus (the median is reported)")
    args = ap.parse_args()

    # stdout carries ONE line, the JSON: libraries that print banners there (RCCL does when a communicator is created) go to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (multi-process GPU work on this pool: the host driver only supports dmabuf IPC)
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
    wd = args.watchdog_seconds if args.watchdog_seconds >= 0 else (1500 if world > 1 else 0)
    if wd:
        import signal

        def _watchdog(signum, frame):
            if rank == 0:
                print(json.dumps({"metric": "bpe_train_throughput", "value": None, "unit": "MB/s", "n_gpus": world, "error": "watchdog: not finished after %d s (rank 0)" % wd}), flush=True)
            log("rank %d: watchdog after %d s -- exiting" % (rank, wd))
            os._exit(3)
        signal.signal(signal.SIGALRM, _watchdog)
        signal.alarm(wd)
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
    pins = json.load(open(PINS)) if os.path.exists(PINS) else {}
    quiet = Quiet()
    tmpdir = tempfile.mkdtemp(prefix="yttm_bench_")
    strong = world > 1 and args.scaling == "strong"
    ctx = dict(L=L, _lib=_lib, torch=torch, np=np, gen=gen, dev=dev, local_rank=local_rank, rank=rank, world=world, dist=dist, args=args,
               pins=pins, quiet=quiet, tmpdir=tmpdir, strong=strong)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    ctx["barrier"] = barrier
    # (YTTM_BENCH_FORCE_COMM=1: a communicator even for one rank -- the N>1 code path of this script and of the library, with every
    # collective of a round, on the one GPU a test box has)
    comm_handle = _init_rccl(L, dist, torch, dev, rank, world, local_rank) if (world > 1 or os.environ.get("YTTM_BENCH_FORCE_COMM")) else None
    ctx["comm"] = comm_handle

    # ---- main workload: configs[1] ------------------------------------------------------------------------------------
    main_res = _bench_train(ctx, args.corpus, args.size_mb, args.steps, args.warmup, measure_touched=(world == 1 and not args.no_touched_pass), keep_host=True)
    out = {
        "metric": "bpe_train_throughput", "value": main_res["value"], "unit": "MB/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "higher_is_better": True,
        "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": main_res["config"], "parity": {}, "roofline": main_res["roofline"], "roofline_pair_count": main_res["roofline_pair_count"],
        "kernels": main_res["kernels"], "phases_s": main_res["phases_s"],
    }
    if main_res.get("hbm_resident"):
        out["value_hbm_resident"] = main_res["hbm_resident"]["value"]
        out["hbm_resident"] = main_res["hbm_resident"]
        out["parity"]["hbm_resident_model_matches_reference"] = main_res["hbm_resident"].pop("_model_ok")
    out["parity"]["corpus_md5_matches"] = main_res["corpus_ok"]
    out["parity"]["model_matches_reference"] = main_res["model_ok"]
    if comm_handle is not None:
        out["rccl_ranks"] = world
        out["config"]["parallelism"] = "one process per GPU; the corpus cut into %d byte ranges at white space; every rank keeps the whole pair table; per round ONE collective: an RCCL all-gather of per-pair delta blocks" % world
    model_path = main_res["model_path"]
    host = main_res.pop("host", None)
    if main_res.get("first_call_s") is not None and not ctx.get("_first_call_done"):
        # what `value` leaves out: the FIRST call of the process -- code objects loaded, ~6 GB of hipMalloc for the pool, the upload
        # workers' pinned chunks and streams created; every later call reuses them.  (HIP itself was initialised by torch before.)
        out.setdefault("e2e", {})["train_first_call"] = {"seconds": round(main_res["first_call_s"], 4), "steady_state_seconds": round(main_res["ms_per_step"] / 1e3, 4),
                                                          "what": "the first yttm_train_bpe_comm(path -> model) of this process on the same file (an untimed warm-up step): cold device-memory pool, pinned chunks, streams, code objects"}

    # ---- encode: configs[3] (and [4]: dropout) with the model just trained --------------------------------------------------
    if not args.no_encode:
        enc = _bench_encode(ctx, model_path, main_res)
        out["encode"] = enc["encode"]
        out["encode_dropout"] = enc["dropout"]
        out["parity"]["encode_fnv_matches"] = enc["fnv_ok"]
        out["parity"]["dropout_distribution_matches"] = enc["dropout_ok"]
        if "e2e" in enc:
            out.setdefault("e2e", {}).update(enc["e2e"])
            # (what compares with the reference's encode_as_ids -- host strings in, host ids out -- is this one, not the device-resident rate)
            out["encode"]["value_host_to_host"] = enc["e2e"]["encode_host_to_host"]["value"]

    # ---- configs[2]: the Zipf corpus -- thousands of short rounds, per-round latency is the whole game ----------------------
    if not args.no_extra and args.corpus == "abcd":
        z = _bench_train(ctx, "zipf", args.size_mb, args.steps, args.warmup, measure_touched=False, keep_host=(world == 1))
        zhost = z.pop("host", None)
        out["extra"] = {"zipf": {"metric": "bpe_train_throughput", "value": z["value"], "unit": "MB/s", "ms_per_step": z["ms_per_step"],
                                 "config": z["config"], "us_per_round": round(z["ms_per_step"] * 1e3 / max(1, z["config"]["merge_rounds"]), 2),
                                 "kernels": z["kernels"], "phases_s": z["phases_s"]}}
        out["parity"]["zipf_corpus_md5_matches"] = z["corpus_ok"]
        out["parity"]["zipf_model_matches_reference"] = z["model_ok"]
        if comm_handle is not None:
            # configs[2] AS WORDED ("8x sharded with RCCL pair-count all-reduce"): the library's own choice for this corpus -- 2.8e6 dedup tokens --
            # is the replicated merge loop (no collective per round; `extra.zipf` above).  The sharded loop is forced here
            # (YTTM_REPLICATE_MAX_TOKENS=0, read when the context is made) so that both are measured side by side.
            os.environ["YTTM_REPLICATE_MAX_TOKENS"] = "0"
            try:
                zs = _bench_train(ctx, "zipf", args.size_mb, args.steps, args.warmup, measure_touched=False, keep_host=False)
            finally:
                del os.environ["YTTM_REPLICATE_MAX_TOKENS"]
            out["extra"]["zipf_sharded"] = {"metric": "bpe_train_throughput", "value": zs["value"], "unit": "MB/s", "ms_per_step": zs["ms_per_step"], "config": zs["config"],
                                            "us_per_round": round(zs["ms_per_step"] * 1e3 / max(1, zs["config"]["merge_rounds"]), 2), "kernels": zs["kernels"],
                                            "phases_s": zs["phases_s"]}
            out["parity"]["zipf_sharded_model_matches_reference"] = zs["model_ok"]
        if world == 1 and not args.no_encode and zhost is not None:
            # natural-language-like sentences: the corpus' own lines (16 Zipf words each) through K5 with the model just trained
            out["extra"]["zipf"]["encode"] = _bench_encode_lines(ctx, z["model_path"], zhost)
        # ---- a large alphabet and an enwik-like word table: train only, pinned against the reference by make_full_pins.py
        if world == 1 and not args.no_extra2:
            for name in ("cjk", "zipf4m"):
                x = _bench_train(ctx, name, args.size_mb, min(args.steps, 2), 1, measure_touched=False, keep_host=False)
                out["extra"][name] = {"metric": "bpe_train_throughput", "value": x["value"], "unit": "MB/s", "ms_per_step": x["ms_per_step"], "config": x["config"],
                                      "us_per_round": round(x["ms_per_step"] * 1e3 / max(1, x["config"]["merge_rounds"]), 2), "kernels": x["kernels"],
                                      "phases_s": x["phases_s"]}
                out["parity"][name + "_corpus_md5_matches"] = x["corpus_ok"]
                out["parity"][name + "_model_matches_reference"] = x["model_ok"]
    else:
        zhost = None

    # ---- beyond 2^32: more than 4 GiB of text; a word seen more than 2^32 times (file -> model, pinned against the reference) ------------
    if world == 1 and not args.no_big and args.corpus == "abcd" and args.size_mb == 1000:
        out.setdefault("extra", {})
        for name in (("c3_8gb", "c8_heavy_word") if args.big_zipf else ("c8_heavy_word",)):
            b = _bench_big(ctx, name)
            if b is not None:
                out["extra"][name] = b
                if isinstance(b.get("chunked_front_end"), dict) and "model_matches_reference" in b["chunked_front_end"]:
                    out["parity"][name + "_chunked_model_matches_reference"] = b["chunked_front_end"]["model_matches_reference"]
                out["parity"][name + "_model_matches_reference"] = b.pop("_model_ok")
                out["parity"][name + "_corpus_md5_matches"] = b.pop("_corpus_ok")

    # ---- CPU baseline: the unmodified reference on this box's host cores (rank 0, N=1 only) --------------------------------
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = _cpu_baseline(ctx, host, zhost, model_path, out)
    shutil.rmtree(tmpdir, ignore_errors=True)
    bad = [k for k, v in out["parity"].items() if v is False]
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        if bad:
            log("PARITY FAILURE:", bad)
        # the LAST line of stderr (a driver that keeps only the tail of the log still shows which pins were checked and how they came out)
        log("parity: " + json.dumps(out["parity"], sort_keys=True))
    if dist is not None:
        dist.destroy_process_group()
    if bad:
        sys.exit(3)


def _make_corpus(ctx, corpus, size_mb):
    """Rank-local shard (host bytes) + what to compare its md5 with.  Strong scaling / single GPU: the pinned file."""
    gen, args, rank, world = ctx["gen"], ctx["args"], ctx["rank"], ctx["world"]
    nbytes = size_mb * 1_000_000
    shared = ctx["strong"] or world == 1
    seed_off = 0 if shared else rank
    if corpus == "abcd":
        host = gen.abcd_corpus(nbytes, seed=19 + seed_off, survey_stream=True)  # SURVEY.md Appendix C gen_abcd: C2 byte for byte
    elif corpus == "cjk":  # large alphabet: 4096 ideographs, space-free clauses (the reference's slowest published cases, benchmark.md:23,39)
        host = gen.cjk_corpus_fast(nbytes, seed=11 + seed_off)
    elif corpus == "zipf4m":  # enwik-like: 4e6-word lexicon, exponent 1.0 (U ~ 2-3e6)
        host = gen.zipf_corpus_fast(nbytes, seed=7 + seed_off, vocab=4_000_000, exponent=1.0)
    else:
        host = gen.zipf_corpus_fast(nbytes, seed=7 + seed_off, vocab=400000)
    pin = ctx["pins"].get(pin_name(corpus, size_mb)) if shared and args.vocab == 32000 else None
    corpus_ok = None
    if pin is not None:
        corpus_ok = hashlib.md5(host).hexdigest() == pin["corpus_md5"]
    total = len(host)
    full = host if shared else None  # (the file of the file -> model steps: one file, every rank reads its byte range of it)
    if ctx["strong"]:
        cuts = split_points(host, world)
        host = host[cuts[rank]:cuts[rank + 1]]
    return host, pin, corpus_ok, total, full


def _bench_train(ctx, corpus, size_mb, steps, warmup, measure_touched, keep_host):
    L, _lib, torch, args = ctx["L"], ctx["_lib"], ctx["torch"], ctx["args"]
    rank, world, dev, local_rank, dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["local_rank"], ctx["dist"]
    t0 = time.time()
    host, pin, corpus_ok, total_bytes, full = _make_corpus(ctx, corpus, size_mb)
    d_corpus = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
    n_local = d_corpus.numel()
    if not ctx["strong"]:
        total_bytes = n_local * world
    tag = os.environ.get("MASTER_PORT", str(os.getpid()))
    model_path = os.path.join(tempfile.gettempdir(), "yttm_bench_%s_%s.model" % (corpus, tag))
    # the corpus FILE of the file -> model steps: written by rank 0, read back once so that it sits in the page cache like the CPU baseline's input
    corpus_path = os.path.join(tempfile.gettempdir(), "yttm_bench_%s_%s.txt" % (corpus, tag)) if full is not None else None
    if corpus_path and rank == 0:
        with open(corpus_path, "wb") as f:
            f.write(full)
    del full
    if corpus_path:
        ctx["barrier"]()
        with open(corpus_path, "rb") as f:
            while f.read(1 << 24):
                pass
    if rank == 0:
        log(f"corpus: {corpus} {n_local/1e6:.1f} MB on this GPU ({total_bytes/1e6:.1f} MB in all) generated+uploaded+written in {time.time()-t0:.1f}s")
    err = C.create_string_buffer(_lib.ERRLEN)
    rep = C.create_string_buffer(16384)
    dev_model_path = model_path + ".hbm"

    def file_step(profile):  # SURVEY.md 8d's metric: yttm_train_bpe_comm(path -> model); with a communicator every rank reads its byte range
        rc = L.yttm_train_bpe_comm(corpus_path.encode(), model_path.encode(), args.vocab, 1.0, 8, 0, 1, 2, 3, local_rank, int(profile), ctx["comm"], rep, 16384, err, _lib.ERRLEN)
        if rc != 0:
            raise RuntimeError("train (file -> model) failed: " + err.value.decode())
        return json.loads(rep.value.decode())

    def train_step(profile, out_model=None):  # the same training with the corpus (this rank's shard) already resident in HBM
        out_model = out_model or dev_model_path
        if ctx["comm"] is not None:
            rc = L.yttm_train_bpe_from_device_comm(C.c_void_p(d_corpus.data_ptr()), n_local, out_model.encode(), args.vocab, 1.0,
                                                   0, 1, 2, 3, local_rank, int(profile), ctx["comm"], rep, 16384, err, _lib.ERRLEN)
        else:
            rc = L.yttm_train_bpe_from_device(C.c_void_p(d_corpus.data_ptr()), n_local, out_model.encode(), args.vocab, 1.0,
                                              0, 1, 2, 3, local_rank, int(profile), rep, 16384, err, _lib.ERRLEN)
        if rc != 0:
            raise RuntimeError("train failed: " + err.value.decode())
        return json.loads(rep.value.decode())

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    timed_step = file_step if corpus_path else train_step  # (--scaling weak: every rank has its own corpus, there is no one file)
    with ctx["quiet"]:
        first_call_s = None
        for i in range(warmup):
            t_first = time.perf_counter()
            timed_step(0)
            if i == 0:
                first_call_s = time.perf_counter() - t_first
        ctx["barrier"]()
        t0 = time.perf_counter()
        reports = [timed_step(1) for _ in range(steps)]
        ctx["barrier"]()
        dt = max_over_ranks(time.perf_counter() - t0)
        hbm = None
        if corpus_path:  # the same steps without the read and the upload: input resident in HBM when the timed region starts
            train_step(0)
            ctx["barrier"]()
            t0 = time.perf_counter()
            dev_reports = [train_step(1) for _ in range(steps)]
            ctx["barrier"]()
            dt_dev = max_over_ranks(time.perf_counter() - t0)
            dr = dev_reports[-1]
            hbm = {"value": round(steps * total_bytes / dt_dev / 1e6, 2), "unit": "MB/s", "ms_per_step": round(dt_dev / steps * 1e3, 2), "steps": steps,
                   "input": "this rank's shard resident in HBM before the timed region (yttm_train_bpe_from_device)",
                   "phases_s": {"frontend": dr["seconds_frontend"], "merge_loop": dr["seconds_merge"], "dump": dr["seconds_io"]},
                   "_model_ok": (md5_file(dev_model_path) == pin["model_md5"]) if (pin is not None and rank == 0) else None}
        touched = None
        if measure_touched:  # measurement pass, outside the timed regions (tiles to the end; the totals also as they stood at the word-mode switch)
            os.environ["YTTM_MEASURE_SPLIT_ROUND"] = str(max(0, int(reports[-1].get("word_switch_round") or 0)))
            try:
                touched = train_step(2)
            finally:
                del os.environ["YTTM_MEASURE_SPLIT_ROUND"]
        ev_rep = None
        if measure_touched and ctx["comm"] is None:  # the same step timed with HIP events around every merge round (not timed)
            os.environ["YTTM_PROFILE_EVENTS"] = "1"
            try:
                ev_rep = train_step(1)
            finally:
                del os.environ["YTTM_PROFILE_EVENTS"]
    value = steps * total_bytes / dt / 1e6
    r = reports[-1]
    model_md5 = md5_file(model_path) if rank == 0 else None
    model_ok = (model_md5 == pin["model_md5"]) if (pin is not None and rank == 0) else None
    if corpus_path and rank == 0:  # (every rank is past the barrier behind its last read)
        os.remove(corpus_path)

    kern = {}
    for name, k in r["kernels"].items():
        if k["launches"] and k["ms"] > 0:
            kern[name] = {"ms_total": round(k["ms"], 3), "launches": k["launches"], "avg_ms": round(k["ms"] / k["launches"], 4),
                          "algorithmic_GB": round(k["bytes"] / 1e9, 4), "GBps": round(k["bytes"] / 1e9 / (k["ms"] / 1e3), 1)}
    traffic, traffic_src = _static_traffic(kern, corpus, size_mb, args, world)
    dom = max(kern, key=lambda n: kern[n]["ms_total"]) if kern else None
    roofline = None
    if dom:
        roofline = {"kernel": dom, "bound": "hbm", "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(kern[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic.get(dom), "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": round(kern[dom]["algorithmic_GB"] * 1e9 / kern[dom]["launches"]),
                    "avg_launch_ms": kern[dom]["avg_ms"], "launches": kern[dom]["launches"],
                    "duration_source": "merge rounds whose candidate scan rides in their last kernel (all but a few of a single-GPU training): the device's 100 MHz clock, "
                                       "first workgroup of the round's first launch -> mailbox published, read from the mailbox every round in the timed region; "
                                       "every other launch: HIP events on the context's stream"}
        ev_ms = None
        if ev_rep is not None and ev_rep["kernels"].get(dom, {}).get("launches"):
            e = ev_rep["kernels"][dom]
            ev_ms = e["ms"]
            roofline["avg_launch_ms_hip_events"] = round(e["ms"] / e["launches"], 4)
            roofline["hip_events_note"] = "one more step outside the timed region with YTTM_PROFILE_EVENTS=1: a hipEventRecord before the round's first launch and one after its last (the interval also holds the launch latency of the first kernel); `achieved` / `frac` use THIS time (it agrees with rocprofv3's kernel durations, profiles/), the device-clock figure of the timed steps is in `device_clock`"
        if dom == "merge_apply":
            # The contract's algorithmic bytes for K4: 8*T_touched + 8*W_touched, "touched" = the WORDS that held a
            # merge site, counted by the untimed measurement pass.  What the kernel actually streams -- every live token once,
            # dirty tiles twice -- is kept beside it as `streamed`.
            streamed = {"bytes_per_launch": roofline["algorithmic_bytes_per_launch"], "GBps": roofline["achieved"], "frac": roofline["frac"],
                        "definition": "4*(live tokens) + 8*(tokens of tiles that held a merge site)"}
            roofline["streamed"] = streamed
            if touched is not None and touched.get("touched_words"):
                b8d = 8 * touched["touched_word_tokens"] + 8 * touched["touched_words"]
                ms_dev = kern[dom]["ms_total"]
                ms_used = ev_ms if ev_ms else ms_dev
                ach = b8d / 1e9 / (ms_used / 1e3)
                roofline.update({"achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                                 "algorithmic_bytes_per_launch": round(b8d / kern[dom]["launches"]),
                                 "algorithmic_bytes_8d": b8d, "frac_8d": round(ach / HBM_PEAK_GBS, 4),
                                 "definition": "8*T_touched + 8*W_touched over all launches / total K4 time; touched = words holding a merge site",
                                 "time_source": "HIP events (one untimed step)" if ev_ms else "device clock of the timed steps",
                                 "device_clock": {"ms_total": ms_dev, "avg_launch_ms": kern[dom]["avg_ms"], "achieved": round(b8d / 1e9 / (ms_dev / 1e3), 1),
                                                  "frac": round(b8d / 1e9 / (ms_dev / 1e3) / HBM_PEAK_GBS, 4),
                                                  "note": "first workgroup of the round's first launch -> mailbox published; leaves out the launch latency"},
                                 "touched_words": touched["touched_words"], "touched_word_tokens": touched["touched_word_tokens"],
                                 "touched_tiles": touched["touched_tiles"], "touched_tile_tokens": touched["touched_tile_tokens"],
                                 "merge_sites": touched["merge_sites"]})
                if traffic.get(dom):
                    roofline["traffic_over_algorithmic"] = round(traffic[dom] / max(1, roofline["algorithmic_bytes_per_launch"]), 2)
                # the two halves of K4 apart: the rounds on tiles (streamed) and the word-mode rounds (work follows the sites)
                sr = touched.get("split_round") or 0
                wl, wms = r.get("merge_launches_word_rounds") or 0, r.get("merge_ms_word_rounds") or 0.0
                if sr and wl and kern[dom]["launches"] > wl:
                    bt = 8 * touched["split_touched_word_tokens"] + 8 * touched["split_touched_words"]
                    bw = b8d - bt
                    tl, tms = kern[dom]["launches"] - wl, ms_dev - wms
                    roofline["halves"] = {
                        "tile_rounds": {"launches": tl, "ms_device_clock": round(tms, 3), "algorithmic_bytes_per_launch": round(bt / tl),
                                        "achieved_GBps": round(bt / 1e9 / (tms / 1e3), 1), "frac": round(bt / 1e9 / (tms / 1e3) / HBM_PEAK_GBS, 4)},
                        "word_mode_rounds": {"launches": wl, "ms_device_clock": round(wms, 3), "algorithmic_bytes_per_launch": round(bw / wl),
                                             "achieved_GBps": round(bw / 1e9 / (wms / 1e3), 1), "frac": round(bw / 1e9 / (wms / 1e3) / HBM_PEAK_GBS, 4)},
                        "note": "contract bytes 8*T_touched + 8*W_touched of the rounds before / from the word-mode switch (round %d), device-clock time of the timed steps" % sr}
    if roofline is not None:
        both_fracs(roofline)
        for hv in (roofline.get("halves") or {}).values():
            if isinstance(hv, dict) and "achieved_GBps" in hv:
                hv["frac_of_measured_copy"] = round(hv["achieved_GBps"] / HBM_COPY_GBS, 4)
        dk = _dominant_rocprof_kernel(corpus, size_mb, args, world, roofline)
        if dk:
            roofline["dominant_rocprof_kernel"] = dk
    roofline_pc = None
    if "pair_count" in kern:
        roofline_pc = {"kernel": "pair_count (K3)", "bound": "hbm", "achieved": kern["pair_count"]["GBps"], "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": round(kern["pair_count"]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic.get("pair_count"),
                       "traffic_source": traffic_src, "algorithmic_bytes_per_launch": round(kern["pair_count"]["algorithmic_GB"] * 1e9),
                       "avg_launch_ms": kern["pair_count"]["avg_ms"]}
        both_fracs(roofline_pc)
    names = {"abcd": "random 'abcd ' corpus (BASELINE.json configs[1])", "zipf": "Zipf ASCII corpus, 400k-word lexicon (BASELINE.json configs[2])",
             "cjk": "CJK-shaped corpus (4096 ideographs, clauses without spaces)", "zipf4m": "Zipf ASCII corpus, 4e6-word lexicon, exponent 1.0 (enwik-like)"}
    cfg = {"workload": f"{size_mb} MB {names[corpus]}, vocab_size={args.vocab}"
                       + (f", one file cut into {world} byte ranges" if ctx['strong'] else (f", {size_mb} MB per GPU" if world > 1 else "")),
           "corpus_bytes_total": total_bytes, "corpus_bytes_this_gpu": n_local, "vocab_size": args.vocab, "unique_words": r["n_unique"],
           "dedup_tokens": r["n_tokens"], "merge_rounds": r["rounds"], "rules": r["rules"],
           "rounds_closed_exhausted": r.get("rounds_exhausted"), "word_mode_from_round": r.get("word_switch_round") or None, "word_mode_rounds": r.get("word_rounds"), "word_mode_one_launch_rounds": r.get("word_fused_rounds"),
           "index_builds": r.get("index_builds"), "batch_splits": r.get("batch_splits"), "batch_extensions": r.get("batch_extensions"),
           "hot_rebuilds": r.get("hot_rebuilds"), "top_refills": r.get("top_refills"), "repacks": r.get("repacks"),
           "input": ("file in the page cache -> yttm_train_bpe_comm(path, model): open + pread into pinned chunks + H2D + train + model file closed (SURVEY.md 8d); "
                     "the HBM-resident figure is `value_hbm_resident`") if corpus_path else "resident in HBM before the timed region (--scaling weak: one corpus per rank, no single file)",
           "front_end_under_the_upload": bool(r.get("front_end_overlapped")),  # (K1, K2a, K2b ran on the parts of the file as they landed: their time is in phases_s.upload)
           "model_md5": model_md5, "pinned_model_md5": pin["model_md5"] if pin else None}
    if ctx["comm"] is not None:
        cfg["multi_gpu_mode"] = ("replicated merge loop: shards gathered once after the local dedup, every rank runs the merge loop alone, no per-round collective"
                                 if r.get("replicated_merge_loop") else "sharded merge loop: every rank applies the batch to its words, per-pair count deltas all-gathered every round")
        cfg["rccl_ranks"] = world
        cfg["exchange_repeats"] = r.get("exchange_retries")
        xk = kern.get("exchange")
        cfg["exchange_us_per_round"] = round(xk["ms_total"] * 1e3 / max(1, xk["launches"]), 2) if xk else None  # (all-gather + fold + the round's scan, device clock)
        k4_ms = kern.get("merge_apply", {}).get("ms_total", 0.0)
        if dist is not None:
            t = torch.zeros(world, dtype=torch.float64, device=dev)
            t[rank] = k4_ms
            dist.all_reduce(t)
            cfg["merge_apply_ms_per_rank"] = [round(float(x), 2) for x in t.tolist()]
        else:
            cfg["merge_apply_ms_per_rank"] = [k4_ms]
    res = {"value": round(value, 2), "ms_per_step": round(dt / steps * 1e3, 2), "config": cfg, "roofline": roofline, "roofline_pair_count": roofline_pc,
           "kernels": kern, "phases_s": {"upload": r.get("seconds_upload", 0.0), "frontend": r["seconds_frontend"], "merge_loop": r["seconds_merge"], "dump": r["seconds_io"]},
           "corpus_ok": corpus_ok, "model_ok": model_ok, "model_path": model_path, "pin": pin, "hbm_resident": hbm, "first_call_s": first_call_s}
    if keep_host:
        res["host"] = host
    del d_corpus
    torch.cuda.empty_cache()
    return res


def _bench_big(ctx, name):
    """A corpus past 2^32 -- bytes (c3_8gb: the Zipf stream at 8 GB) or occurrences of one word (c8_heavy_word: 'a' 4.4e9 times; the reference
    counts word frequencies in uint64, bpe.cpp:382-385) -- streamed to a file (never whole in host memory), trained file -> model, the
    model's md5 against the unmodified reference's (tests/golden/full_size_pins.json, made by tests/golden/make_full_pins.py)."""
    L, _lib, gen, args = ctx["L"], ctx["_lib"], ctx["gen"], ctx["args"]
    pin = ctx["pins"].get(name)
    if pin is None or args.vocab != 32000:
        return None
    tmp = tempfile.gettempdir()
    if shutil.disk_usage(tmp).free < pin["corpus_bytes"] + (2 << 30):
        return {"skipped": "not enough room in %s for the %.1f GB corpus file" % (tmp, pin["corpus_bytes"] / 1e9), "_model_ok": None, "_corpus_ok": None}
    path = os.path.join(tmp, "yttm_bench_%s_%d.txt" % (name, os.getpid()))
    model = path + ".model"
    t0 = time.time()
    chunks = gen._zipf_chunks(8_000_000_000, seed=7, vocab=400000) if name == "c3_8gb" else gen.heavy_word_chunks(4_400_000_000)
    nbytes, md5 = gen.stream_to_file(path, chunks)
    t_gen = time.time() - t0
    err = C.create_string_buffer(_lib.ERRLEN)
    rep = C.create_string_buffer(16384)
    times = []
    try:
        with ctx["quiet"]:
            for _ in range(2):  # (the file is in the page cache: it was just written)
                t0 = time.perf_counter()
                rc = L.yttm_train_bpe_comm(path.encode(), model.encode(), args.vocab, 1.0, 8, 0, 1, 2, 3, ctx["local_rank"], 1, None, rep, 16384, err, _lib.ERRLEN)
                times.append(time.perf_counter() - t0)
                if rc != 0:
                    return {"error": err.value.decode(), "corpus_bytes": nbytes, "_model_ok": False, "_corpus_ok": md5 == pin["corpus_md5"]}
        r = json.loads(rep.value.decode())
        best = min(times)
        # The same file with the front end forced into chunks of 512 MB (VERDICT r4 "missing" #2: corpora beyond ~HBM/6): the text crosses
        # the device chunk by chunk, only the distinct words' bytes stay; same model, and the pool's high-water mark says what it took.
        chunked = None
        os.environ["YTTM_FE_CHUNK_MB"] = "512"
        try:
            tcs = {}
            with ctx["quiet"]:
                for variant in ("serial", "overlapped", "overlapped"):  # (serial: upload a chunk, then work on it -- round 5's first version)
                    if variant == "serial":
                        os.environ["YTTM_FE_CHUNK_SERIAL"] = "1"
                    else:
                        os.environ.pop("YTTM_FE_CHUNK_SERIAL", None)
                    t0 = time.perf_counter()
                    rc = L.yttm_train_bpe_comm(path.encode(), (model + ".chunked").encode(), args.vocab, 1.0, 8, 0, 1, 2, 3, ctx["local_rank"], 1, None, rep, 16384, err, _lib.ERRLEN)
                    tcs[variant] = min(tcs.get(variant, 1e9), time.perf_counter() - t0)
                    if rc != 0:
                        break
            dtc = tcs.get("overlapped", 0.0)
            if rc == 0:
                rc_ = json.loads(rep.value.decode())
                chunked = {"chunk_MB": 512, "chunks": rc_["front_end_chunks"], "seconds": round(dtc, 4), "MBps": round(nbytes / 1e6 / dtc, 1),
                           "seconds_upload_then_work": round(tcs["serial"], 4),
                           "peak_device_GB": round(rc_["peak_device_bytes"] / 1e9, 3), "peak_device_GB_whole_text": round(r["peak_device_bytes"] / 1e9, 3),
                           "model_matches_reference": md5_file(model + ".chunked") == pin["model_md5"]}
            else:
                chunked = {"error": err.value.decode()}
        finally:
            del os.environ["YTTM_FE_CHUNK_MB"]
            os.environ.pop("YTTM_FE_CHUNK_SERIAL", None)
            if os.path.exists(model + ".chunked"):
                os.remove(model + ".chunked")
        return {"metric": "bpe_train_throughput", "value": round(nbytes / 1e6 / best, 2), "unit": "MB/s", "seconds": round(best, 4), "all_seconds": [round(t, 4) for t in times],
                "chunked_front_end": chunked,
                "corpus": pin["corpus"], "corpus_bytes": nbytes, "corpus_generation_seconds": round(t_gen, 1), "unique_words": r["n_unique"], "dedup_tokens": r["n_tokens"],
                "merge_rounds": r["rounds"], "rules": r["rules"],
                "phases_s": {"upload": r["seconds_upload"], "frontend": r["seconds_frontend"], "merge_loop": r["seconds_merge"], "dump": r["seconds_io"]},
                "kernels_ms": {k: round(v["ms"], 3) for k, v in r["kernels"].items() if v["launches"]},
                "input": "file in the page cache -> yttm_train_bpe_comm(path, model), best of 2", "model_md5": md5_file(model), "pinned_model_md5": pin["model_md5"],
                "reference_seconds_build_container": pin.get("reference_train_seconds_build_container"),
                "_model_ok": md5_file(model) == pin["model_md5"], "_corpus_ok": md5 == pin["corpus_md5"]}
    finally:
        for f in (path, model):
            if os.path.exists(f):
                os.remove(f)


def _static_traffic(kern, corpus, size_mb, args, world):
    """HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2-corrected + WRITE_SIZE, separate --pmc
    runs of this same command; tools/pmc_summary.py).  STATIC: read from profiles/, not measured by this run."""
    traffic = {}
    tag = {"abcd": "1gb", "zipf": "zipf", "cjk": "cjk"}.get(corpus)
    pmc_file = profile_for(tag, "pmc_hbm.json") if tag else None
    if not pmc_file or not os.path.exists(pmc_file):
        return traffic, None
    if not (size_mb == 1000 and args.vocab == 32000 and world == 1):
        return traffic, None
    pm = json.load(open(pmc_file))
    # A profile is quoted only for the build it was taken of (the r4 file predated three commits of the head it was quoted at).
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_summary import source_sha16
    have, want = (pm.get("_meta") or {}).get("source_sha16"), source_sha16(ROOT)
    if have != want:
        return traffic, "none: profiles/%s was taken of sources %s, this build is %s -- re-run tools/profile_round.sh" % (os.path.basename(pmc_file), have, want)
    pm = {k: v for k, v in pm.items() if not k.startswith("_")}

    def family(k):  # k_tiles<SLOT, WPB, MERGE, LDSR>: MERGE=false is K3 (pair count), true is K4 (merge apply)
        if k.startswith("k_tiles<"):
            return "merge_apply" if k[len("k_tiles<"):].split(", ")[2] == "true" else "pair_count"
        for name, prefixes in (("char_hist", ("k_scan_bytes<0>",)), ("segments", ("k_scan_bytes<1>",)), ("dedup", ("k2b_insert_words",)),
                               ("merge_apply", ("k_giant<true", "k_words<", "k_delta_apply", "k_wgather", "k_round_begin")),
                               ("pair_count", ("k_giant<false", "k_pair_count_dense")),
                               ("cand_scan", ("k_hot_scan", "k_cand_scan", "k_top_scan", "k_top_rebuild", "k_hot_rebuild", "k_idx_", "k_words_init"))):
            if k.startswith(prefixes):
                return name
        return None
    # (the profiled command trains more than once per process -- the file step, the HBM-resident warm-up and step: the histogram is compacted
    # once per training; K1 itself runs once per part of the text in a file step, gpu_ctx.cpp upload_overlapped)
    n_train = max(1, sum(pm[k]["launches"] for k in pm if k.startswith("k_hist_compact")))
    for name in ("char_hist", "segments", "dedup", "pair_count", "merge_apply", "cand_scan"):
        ks = [k for k in pm if family(k) == name]
        if ks and name in kern:
            total = sum(pm[k]["traffic_bytes_per_launch"] * pm[k]["launches"] for k in ks) / n_train
            traffic[name] = round(total / max(1, kern[name]["launches"]))
    return traffic, "static: bytes per launch from profiles/%s (rocprofv3 --pmc passes of this command, tools/profile_round.sh; a PMC pass cannot share a run with the timed region, so it is not re-measured here)" % os.path.basename(pmc_file)


def _dominant_rocprof_kernel(corpus, size_mb, args, world, roofline):
    """The kernel rocprofv3 ranks first among the training's kernels, BY NAME (`roofline.kernel = merge_apply` is a family of four
    rocprof kernels), with its own bytes, time and fraction so that the figure can be checked against profiles/*_kernel_stats.csv without
    arithmetic.  STATIC like `traffic`: calls and average duration come from the committed --kernel-trace --stats summary of this command
    (quoted only when the PMC summary taken in the same profile_round.sh run names this build's sources); the bytes are this run's -- the
    contract's 8*T_touched + 8*W_touched of the rounds that kernel serves (word-mode rounds for k_words, tile rounds for k_tiles)."""
    import csv
    tag = {"abcd": "1gb", "zipf": "zipf", "cjk": "cjk"}.get(corpus)
    if not tag or not (size_mb == 1000 and args.vocab == 32000 and world == 1):
        return None
    stats, pmc = profile_for(tag, "kernel_stats.csv"), profile_for(tag, "pmc_hbm.json")
    if not stats or not pmc or os.path.basename(stats).split("_")[0] != os.path.basename(pmc).split("_")[0]:
        return None
    try:
        meta = (json.load(open(pmc)).get("_meta") or {})
        if meta.get("source_sha16") != sources_sha():
            return {"note": "none: profiles/%s is of sources %s, this build is %s -- re-run tools/profile_round.sh" % (os.path.basename(stats), meta.get("source_sha16"), sources_sha())}
        rows = [r for r in csv.DictReader(open(stats)) if r.get("kernel")]
    except Exception as e:  # noqa: BLE001
        return {"note": "unreadable profile: %s" % e}
    train = [r for r in rows if r["kernel"].startswith(("k_words<", "k_tiles<", "k_giant<", "k_delta_apply", "k_wgather", "k_idx_", "k_hot_", "k_top_", "k_cand_", "k_pair_count"))]
    if not train:
        return None
    top = max(train, key=lambda r: float(r["total_ms"]))
    n_train = max(1, sum(int(r["calls"]) for r in rows if r["kernel"].startswith("k_hist_compact")))
    halves = roofline.get("halves") or {}
    half = halves.get("word_mode_rounds") if top["kernel"].startswith("k_words<") else halves.get("tile_rounds") if top["kernel"].startswith("k_tiles<") else None
    out = {"name": top["kernel"], "calls_per_training": round(int(top["calls"]) / n_train, 1), "avg_us": float(top["avg_us"]), "pct_of_gpu_time": float(top["pct"]),
           "source": "profiles/%s (rocprofv3 --kernel-trace --stats of this command, %d trainings in the process)" % (os.path.basename(stats), n_train)}
    if half:
        b = half["algorithmic_bytes_per_launch"]
        ach = b / 1e9 / (float(top["avg_us"]) / 1e6)
        out.update({"algorithmic_bytes_per_launch": b, "achieved": round(ach, 1), "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "frac_of_measured_copy": round(ach / HBM_COPY_GBS, 4),
                    "bytes_definition": "8*T_touched + 8*W_touched of the rounds this kernel serves (this run's measurement pass) / their launches"})
    return out


def _encode_traffic(args, world, cached):
    """HBM traffic of configs[3]'s encode kernels per batch from the committed PMC passes (profiles/rN_encode10m_pmc_hbm.json), quoted only
    for this build's sources.  Sum over the kernels of the path that ran (word cache: insert, list, words, count, scatter; else k5_encode)."""
    f = profile_for("encode10m", "pmc_hbm.json")
    if not f or world != 1 or args.encode_sentences != 10_000_000:
        return None, None
    pm = json.load(open(f))
    if (pm.get("_meta") or {}).get("source_sha16") != sources_sha():
        return None, "none: profiles/%s is of other sources than this build" % os.path.basename(f)
    names = ("k5w_", "k5_words", "k5_gather") if cached else ("k5_encode<false>", "k5_gather")
    per = {k: v["traffic_bytes_per_launch"] for k, v in pm.items() if not k.startswith("_") and k.startswith(names)}
    return (sum(per.values()) if per else None), "static: profiles/%s, per batch of 10^7 sentences: %s" % (os.path.basename(f), per)


def _bench_encode(ctx, model_path, main_res):
    L, _lib, torch, np, gen, args = ctx["L"], ctx["_lib"], ctx["torch"], ctx["np"], ctx["gen"], ctx["args"]
    rank, world, dev, local_rank, dist = ctx["rank"], ctx["world"], ctx["dev"], ctx["local_rank"], ctx["dist"]
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
    steps = max(1, args.steps)

    def run(dropout):
        def step():
            rc = L.yttm_encode_device(h, C.c_void_p(d_bytes.data_ptr()), C.c_void_p(d_off.data_ptr()), n_sent, d_bytes.numel(),
                                      line + 1, 0, 0, 0, dropout, C.byref(n_ids), C.byref(kms), err, _lib.ERRLEN)
            if rc != 0:
                raise RuntimeError(err.value.decode())
        step()
        ctx["barrier"]()
        t0 = time.perf_counter()
        k_ms = []
        for _ in range(steps):
            step()
            k_ms.append(kms.value)
        ctx["barrier"]()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        sps = steps * n_sent * world / dt
        alg_bytes = d_bytes.numel() + 8 * (n_sent + 1) * 2 + 4 * n_ids.value  # B_in + 16(S+1) + 4 K_out
        kavg = sum(k_ms) / len(k_ms)
        return {"metric": "encode_sentences_per_s", "value": round(sps, 1), "unit": "sentences/s", "sentences_per_gpu": n_sent,
                "sentence_chars": line, "dropout_prob": dropout, "ids_per_sentence": round(n_ids.value / n_sent, 3),
                "ms_per_step": round(dt / steps * 1e3, 2), "kernel_ms": round(kavg, 3), "input": "sentences and offsets resident in HBM, ids left in HBM",
                "roofline": {"kernel": "k5_encode", "bound": "hbm", "achieved": round(alg_bytes / 1e9 / (kavg / 1e3), 1), "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": round(alg_bytes / 1e9 / (kavg / 1e3) / HBM_PEAK_GBS, 4), "traffic": None,
                             "algorithmic_bytes_per_launch": alg_bytes}}

    res = {"dropout": run(0.1)}  # configs[4]: per-lane RNG, so parity is a distribution match against the reference (n_threads=1) -- below
    res["dropout_ok"] = None
    c5 = os.path.join(ROOT, "tests", "golden", "c5_dropout_pin.json")
    if rank == 0 and os.path.exists(c5) and main_res["pin"] is not None and main_res["model_ok"]:
        pin5 = json.load(open(c5))
        m5 = pin5["sentences"]
        if pin5["model_md5"] == main_res["pin"]["model_md5"] and n_sent >= m5 and hashlib.md5(host[: m5 * (line + 1)]).hexdigest() == pin5["input_md5_first_1m"]:
            ids = np.zeros(n_ids.value, dtype=np.int32)
            off = np.zeros(n_sent + 1, dtype=np.uint64)
            L.yttm_encode_fetch(h, ids.ctypes.data_as(_lib.i32p), off.ctypes.data_as(_lib.u64p), n_sent, err, _lib.ERRLEN)
            res["dropout"]["distribution"] = _dropout_distribution(np, ids, off, m5, pin5)
            res["dropout_ok"] = res["dropout"]["distribution"]["matches"]
            del ids, off
    L.yttm_encoder_set_cache(h, 0, 0)
    direct = run(0.0)            # every word occurrence through K5, as the reference does
    L.yttm_encoder_set_cache(h, 2, 8 << 20)
    res["encode"] = run(0.0)     # configs[3], the library's default path (word cache); last: its ids are the ones checked below
    words = int(L.yttm_encode_cache_words(h))
    res["encode"]["word_cache"] = {"distinct_words": words, "without_cache_sentences_per_s": direct["value"], "without_cache_kernel_ms": direct["kernel_ms"],
                                   "gain": round(res["encode"]["value"] / direct["value"], 3)}
    res["encode"]["roofline"]["kernel"] = "k5w_insert + k5_words (the distinct words) + k5w_count + k5w_scatter" if words else "k5_encode"
    tr, src = _encode_traffic(args, world, bool(words))
    res["encode"]["roofline"]["traffic"] = tr
    res["encode"]["roofline"]["traffic_source"] = src
    if tr:
        res["encode"]["roofline"]["traffic_over_algorithmic"] = round(tr / res["encode"]["roofline"]["algorithmic_bytes_per_launch"], 2)
    both_fracs(res["encode"]["roofline"])
    both_fracs(res["dropout"]["roofline"])
    res["dropout"]["compared"] = "distribution match on the first 1 000 000 of the 10 M sentences (the reference at n_threads=1 needs ~40 s per million); every sentence is encoded and timed"
    # ---- parity: FNV-1a-64 of (len, ids...) per sentence over ALL sentences vs the reference's (pinned) ---------------------
    ids = np.zeros(n_ids.value, dtype=np.int32)
    off = np.zeros(n_sent + 1, dtype=np.uint64)
    L.yttm_encode_fetch(h, ids.ctypes.data_as(_lib.i32p), off.ctypes.data_as(_lib.u64p), n_sent, err, _lib.ERRLEN)
    pin = ctx["pins"].get("c4_10m")
    fnv_ok = None
    if rank == 0 and pin and main_res["pin"] is not None and main_res["pin"]["model_md5"] == pin["model_md5"] and main_res["model_ok"]:
        if n_sent == pin["n_sentences"]:
            got = "%016x" % L.yttm_ids_fnv1a64(ids.ctypes.data_as(_lib.i32p), off.ctypes.data_as(_lib.u64p), n_sent)
            fnv_ok = (got == pin["fnv1a64"]) and int(n_ids.value) == pin["n_ids"]
            res["encode"]["fnv1a64"] = got
            res["encode"]["compared"] = "all %d sentences against the reference's encode_as_ids (tests/golden/full_size_pins.json c4_10m)" % n_sent
        elif n_sent >= 1_000_000:
            m = 1_000_000
            got = "%016x" % L.yttm_ids_fnv1a64(ids.ctypes.data_as(_lib.i32p), off.ctypes.data_as(_lib.u64p), m)
            fnv_ok = got == pin["first_1m"]["fnv1a64"]
            res["encode"]["compared"] = "first 1 000 000 sentences against the reference (pin c4_10m.first_1m)"
    res["fnv_ok"] = fnv_ok
    del ids, off
    # ---- host -> host through the drop-in call, and the Python list API (yttm.pyx:87-124) ------------------------------------
    if world == 1 and not args.no_e2e:
        h_off = (np.arange(n_sent + 1, dtype=np.uint64) * (line + 1))
        blob = host
        p_ids, p_off = _lib.i32p(), _lib.u64p()
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            rc = L.yttm_encode_as_ids(h, blob, h_off.ctypes.data_as(_lib.u64p), n_sent, 0, 0, 0, 0.0, C.byref(p_ids), C.byref(p_off), err, _lib.ERRLEN)
            dt = time.perf_counter() - t0
            if rc != 0:
                raise RuntimeError(err.value.decode())
            L.yttm_free(C.cast(p_ids, C.c_void_p))
            L.yttm_free(C.cast(p_off, C.c_void_p))
            best = dt if best is None else min(best, dt)
        res["e2e"] = {"encode_host_to_host": {"value": round(n_sent / best, 1), "unit": "sentences/s", "seconds": round(best, 4),
                                              "what": "yttm_encode_as_ids: packed host bytes+offsets -> H2D -> K5 -> D2H -> malloc'ed host ids; best of 2"}}
        import youtokentome_amd as yttm
        bpe = yttm.BPE(model_path)
        m = min(n_sent, 1_000_000)
        sents = host[: m * (line + 1)].decode().split("\n")[:m]
        bpe.encode(sents[:1000])
        t0 = time.perf_counter()
        got = bpe.encode(sents, output_type=yttm.OutputType.ID)
        dt = time.perf_counter() - t0
        res["e2e"]["encode_python_list_api"] = {"value": round(m / dt, 1), "unit": "sentences/s", "sentences": m, "seconds": round(dt, 4),
                                                "what": "youtokentome_amd.BPE.encode(list[str]) -> list[list[int]] (yttm.pyx:87-109 boundary)",
                                                "ids": sum(len(s) for s in got)}
        del got, sents
    L.yttm_encoder_destroy(h)
    res["_host"] = host
    ctx["_enc_host"] = host
    del d_bytes, d_off
    torch.cuda.empty_cache()
    return res


def _dropout_distribution(np, ids, off, m, pin):
    """BASELINE.json configs[4] / SURVEY.md 8d C5: the GPU's BPE-dropout output against the reference's (unmodified, n_threads=1, fresh
    process; tests/golden/c5_dropout_pin.json, made by tests/golden/make_full_pins.py c5) on the SAME first `m` sentences: mean ids per
    sentence within 1 %, two-sample KS on the sentence lengths (alpha ~ 0.001), chi-square per degree of freedom on the unigram id
    counts (bins with >= 20 counts).  The rest of the GPU's 10 M sentences is reported too (same process, more text)."""
    lens_all = np.diff(off.astype(np.int64))
    lens = lens_all[:m]
    n_ids_m = int(off[m])
    hw_len = np.asarray(pin["len_hist"], dtype=np.float64)
    hg_len = np.bincount(lens, minlength=len(hw_len)).astype(np.float64)
    k = max(len(hw_len), len(hg_len))
    hw_len = np.pad(hw_len, (0, k - len(hw_len)))
    hg_len = np.pad(hg_len, (0, k - len(hg_len)))
    ks = float(np.abs(np.cumsum(hw_len) / hw_len.sum() - np.cumsum(hg_len) / hg_len.sum()).max())
    ks_crit = 1.95 * float(np.sqrt((hw_len.sum() + hg_len.sum()) / (hw_len.sum() * hg_len.sum())))
    hw = np.asarray(pin["id_hist"], dtype=np.float64)
    hg = np.bincount(ids[:n_ids_m], minlength=len(hw)).astype(np.float64)[: len(hw)]
    mask = (hw + hg) >= 20
    chi = float((((hg[mask] - hw[mask]) ** 2) / (hg[mask] + hw[mask])).sum() / max(1, int(mask.sum()) - 1))
    mean_g, mean_w = n_ids_m / m, pin["ids"] / pin["sentences"]
    ok = abs(mean_g - mean_w) / mean_w < 0.01 and ks < ks_crit and chi < 1.5
    return {"matches": bool(ok), "sentences_compared": m, "ids_per_sentence_gpu": round(mean_g, 4), "ids_per_sentence_reference": round(mean_w, 4),
            "ks_sentence_lengths": round(ks, 6), "ks_critical_alpha_0.001": round(ks_crit, 6), "chi2_per_dof_unigram_ids": round(chi, 4),
            "chi2_bins": int(mask.sum()), "chi2_limit": 1.5, "ids_per_sentence_gpu_all": round(float(lens_all.mean()), 4),
            "reference": pin["reference"]}


def _bench_encode_lines(ctx, model_path, text):
    """Device-resident encode of the newline-separated lines of `text` (bytes); FNV of the result for the on-box comparison with the reference."""
    L, _lib, torch, np, args = ctx["L"], ctx["_lib"], ctx["torch"], ctx["np"], ctx["args"]
    dev, local_rank = ctx["dev"], ctx["local_rank"]
    arr = np.frombuffer(text, dtype=np.uint8)
    ends = np.flatnonzero(arr == 10).astype(np.int64)
    n_sent = len(ends)
    # sentence i = bytes [off[i], off[i+1]): the newline stays at the end of its line (white space, like in the abcd stream)
    off = np.zeros(n_sent + 1, dtype=np.int64)
    off[1:] = ends + 1
    d_bytes = torch.frombuffer(bytearray(text[: int(off[-1])]), dtype=torch.uint8).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    max_len = int((off[1:] - off[:-1]).max())
    err = C.create_string_buffer(_lib.ERRLEN)
    h = C.c_void_p()
    if L.yttm_encoder_create(model_path.encode(), 1, local_rank, C.byref(h), err, _lib.ERRLEN) != 0:
        raise RuntimeError(err.value.decode())
    n_ids, kms = C.c_uint64(), C.c_double()

    def step():
        rc = L.yttm_encode_device(h, C.c_void_p(d_bytes.data_ptr()), C.c_void_p(d_off.data_ptr()), n_sent, d_bytes.numel(), max_len, 0, 0, 0, 0.0,
                                  C.byref(n_ids), C.byref(kms), err, _lib.ERRLEN)
        if rc != 0:
            raise RuntimeError(err.value.decode())
    steps = max(1, args.steps)

    def timed():
        step()
        ctx["barrier"]()
        t0 = time.perf_counter()
        k_ms = []
        for _ in range(steps):
            step()
            k_ms.append(kms.value)
        ctx["barrier"]()
        return time.perf_counter() - t0, k_ms
    L.yttm_encoder_set_cache(h, 0, 0)  # every word occurrence through K5
    dt_direct, k_direct = timed()
    L.yttm_encoder_set_cache(h, 2, 8 << 20)  # the default: distinct words once
    dt, k_ms = timed()
    words = int(L.yttm_encode_cache_words(h))
    ids = np.zeros(n_ids.value, dtype=np.int32)
    o64 = np.zeros(n_sent + 1, dtype=np.uint64)
    L.yttm_encode_fetch(h, ids.ctypes.data_as(_lib.i32p), o64.ctypes.data_as(_lib.u64p), n_sent, err, _lib.ERRLEN)
    fnv = "%016x" % L.yttm_ids_fnv1a64(ids.ctypes.data_as(_lib.i32p), o64.ctypes.data_as(_lib.u64p), n_sent)
    L.yttm_encoder_destroy(h)
    del d_bytes, d_off
    torch.cuda.empty_cache()
    return {"metric": "encode_sentences_per_s", "value": round(steps * n_sent / dt, 1), "unit": "sentences/s", "sentences": n_sent,
            "mean_sentence_bytes": round(float(off[-1]) / n_sent, 1), "ids_per_sentence": round(n_ids.value / n_sent, 3),
            "kernel_ms": round(sum(k_ms) / len(k_ms), 3), "bytes_per_s": round(steps * float(off[-1]) / dt, 1), "fnv1a64": fnv,
            "word_cache": {"distinct_words": words, "without_cache_sentences_per_s": round(steps * n_sent / dt_direct, 1),
                           "without_cache_kernel_ms": round(sum(k_direct) / len(k_direct), 3), "gain": round(dt_direct / dt, 3)},
            "input": "the lines of the Zipf corpus, resident in HBM; ids left in HBM"}


def _taskset():
    """Pin the CPU baseline to 8 cores when taskset and >= 8 cores exist."""
    n = os.cpu_count() or 1
    if shutil.which("taskset") and n >= 8:
        return ["taskset", "-c", "0-7"], 8
    return [], min(8, n)


def _cpu_baseline(ctx, host, zhost, model_path, out):
    args = ctx["args"]
    ref = os.path.join(ROOT, "oracle", "_ref", "yttm_ref_prod")
    if not os.path.exists(ref):
        return {"value": None, "unit": "MB/s", "cores": 0, "kind": "reference", "sample": "oracle/_ref/yttm_ref_prod not built"}
    pre, cores = _taskset()
    tmpdir = ctx["tmpdir"]
    os.makedirs(tmpdir, exist_ok=True)

    runs = max(1, args.cpu_runs)

    def train(buf, tag):
        sample = buf
        if args.cpu_sample_mb:
            sample = buf[: args.cpu_sample_mb * 1_000_000]
            sample = sample[: sample.rfind(b"\n") + 1]
        path = os.path.join(tmpdir, tag + ".txt")
        with open(path, "wb") as f:
            f.write(sample)
        model = os.path.join(tmpdir, tag + ".model")
        secs = []
        for _ in range(runs):  # 3 runs, median, page cache warm
            r = subprocess.run(pre + [ref, "train", path, model, str(args.vocab), "1.0", "8", "0", "1", "2", "3"], capture_output=True, text=True)
            secs.append(json.loads(r.stdout.strip().splitlines()[-1])["train_seconds"])
        os.remove(path)
        secs.sort()
        return len(sample), secs[len(secs) // 2], secs

    res = {"value": None, "unit": "MB/s", "cores": cores, "kind": "reference", "pinned": bool(pre)}
    try:
        n, secs, all_secs = train(host, "cpu_c2")
        res["value"] = round(n / 1e6 / secs, 2)
        res["train_seconds"] = round(secs, 2)
        res["all_train_seconds"] = [round(x, 2) for x in all_secs]
        res["sample"] = (f"unmodified reference (oracle/_ref/yttm_ref_prod = bpe.cpp as shipped, -O3), n_threads=8"
                         f"{', taskset -c 0-7' if pre else ''}, train_bpe (C++ boundary: file -> model) on "
                         f"{'the SAME full' if not args.cpu_sample_mb else 'the first'} {n/1e6:.0f} MB of the corpus, vocab {args.vocab}, median of {runs} runs")
        res["gpu_over_cpu"] = round(out["value"] / res["value"], 1)  # file -> model on both sides
        if out.get("value_hbm_resident"):
            res["gpu_hbm_resident_over_cpu"] = round(out["value_hbm_resident"] / res["value"], 1)
    except Exception as e:  # noqa: BLE001
        res["error"] = str(e)
    det = os.path.join(ROOT, "oracle", "_ref", "yttm_ref_det")
    if os.path.exists(det) and res.get("value"):  # "prod-oracle and det-oracle, same files, same box" -- the parity target's own time, one run
        try:
            prod_ref, ref, runs_saved = ref, det, runs
            runs = 1
            n, secs, _ = train(host, "cpu_c2_det")
            res["det_oracle"] = {"value": round(n / 1e6 / secs, 2), "unit": "MB/s", "train_seconds": round(secs, 2), "cores": cores, "runs": 1,
                                 "sample": "the same reference sources with -DDETERMINISTIC_QUEUE (oracle/_ref/yttm_ref_det: the build whose model is the parity target), same file, same flags"}
        except Exception as e:  # noqa: BLE001
            res["det_oracle"] = {"error": str(e)}
        finally:
            ref, runs = prod_ref, runs_saved
    if zhost is not None:
        try:
            n, secs, all_secs = train(zhost, "cpu_c3")
            res["zipf"] = {"value": round(n / 1e6 / secs, 2), "unit": "MB/s", "train_seconds": round(secs, 2), "all_train_seconds": [round(x, 2) for x in all_secs], "cores": cores,
                           "sample": f"same reference build and flags on the full {n/1e6:.0f} MB Zipf corpus (configs[2])"}
        except Exception as e:  # noqa: BLE001
            res["zipf"] = {"error": str(e)}
    enc_host = ctx.get("_enc_host")
    if enc_host is not None and "encode" in out:
        lines = os.path.join(tmpdir, "enc.txt")
        with open(lines, "wb") as f:
            f.write(enc_host)
        for key, dropout, nmax in (("encode", "0.0", -1), ("encode_dropout", "0.1", 2_000_000)):
            try:
                r = subprocess.run(pre + [ref, "encode_bench", model_path, lines, "8", dropout, str(nmax)], capture_output=True, text=True)
                j = json.loads(r.stdout.strip().splitlines()[-1])
                res[key] = {"value": round(j["sentences"] / j["encode_seconds"], 1), "unit": "sentences/s", "cores": cores,
                            "sample": f"{j['sentences']} sentences of 128 chars, encode_as_ids, n_threads=8, dropout {dropout}",
                            "ids_per_sentence": round(j["ids"] / j["sentences"], 3)}
                if key == "encode":
                    res[key]["fnv1a64"] = j["fnv1a64"]
                    res[key]["ids_match_gpu"] = (j["fnv1a64"] == out["encode"].get("fnv1a64")) if out["encode"].get("fnv1a64") else None
            except Exception as e:  # noqa: BLE001
                res[key] = {"error": str(e)}
        os.remove(lines)
    zenc = out.get("extra", {}).get("zipf", {}).get("encode")
    if zhost is not None and zenc is not None:
        lines = os.path.join(tmpdir, "zenc.txt")
        with open(lines, "wb") as f:
            f.write(zhost)
        try:
            zmodel = os.path.join(tempfile.gettempdir(), "yttm_bench_zipf_%s.model" % os.environ.get("MASTER_PORT", str(os.getpid())))
            r = subprocess.run(pre + [ref, "encode_bench", zmodel, lines, "8", "0.0", "-1"], capture_output=True, text=True)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            res.setdefault("zipf", {})["encode"] = {"value": round(j["sentences"] / j["encode_seconds"], 1), "unit": "sentences/s", "cores": cores,
                                                    "sample": f"{j['sentences']} lines of the Zipf corpus, encode_as_ids, n_threads=8",
                                                    "fnv1a64": j["fnv1a64"], "ids_match_gpu": j["fnv1a64"] == zenc["fnv1a64"]}
            out["parity"]["zipf_encode_fnv_matches_reference_on_this_box"] = res["zipf"]["encode"]["ids_match_gpu"]
        except Exception as e:  # noqa: BLE001
            res.setdefault("zipf", {})["encode"] = {"error": str(e)}
        os.remove(lines)
    res["python_boundary"] = _python_boundary(ctx, pre, cores, model_path, enc_host, zhost, out, host)
    return res


def _python_boundary(ctx, pre, cores, model_path, enc_host, zhost, out, host=None):
    """The reference's OWN Python boundary and command line on this box (its Cython module, compiled from the unmodified sources by
    oracle/Makefile into oracle/_ref/pyref, travels with the tree): youtokentome.BPE.encode(list[str]) (yttm.pyx:87-109) and
    `yttm encode` stdin -> stdout (bpe.cpp:1942-2014, what benchmark.md's "Tokenization" times), beside the drop-in's through shim/."""
    pyref = os.path.join(ROOT, "oracle", "_ref", "pyref")
    if not os.path.exists(os.path.join(pyref, "_youtokentome_cython.so")):
        return {"error": "oracle/_ref/pyref not built (make -C oracle ref, where /root/reference exists)"}
    tmpdir = ctx["tmpdir"]
    res = {}
    script = os.path.join(ROOT, "tools", "python_boundary.py")
    shim = os.path.join(ROOT, "shim")

    def env_for(which):
        e = dict(os.environ)
        e["PYTHONPATH"] = (pyref if which == "reference" else shim + os.pathsep + ROOT) + os.pathsep + e.get("PYTHONPATH", "")
        return e
    if enc_host is not None:
        lines = os.path.join(tmpdir, "pb.txt")
        m = 1_000_000
        with open(lines, "wb") as f:
            f.write(enc_host[: m * 129])
        for which in ("reference", "drop_in"):
            try:
                r = subprocess.run((pre if which == "reference" else []) + [sys.executable, script, model_path, lines, str(m), "8", "3"],
                                   capture_output=True, text=True, env=env_for(which))
                j = json.loads(r.stdout.strip().splitlines()[-1])
                res["encode_list_" + which] = {"value": round(j["sentences"] / j["median_seconds"], 1), "unit": "sentences/s", "sentences": j["sentences"],
                                               "ids": j["ids"], "seconds": [round(x, 3) for x in j["seconds"]], "module": os.path.relpath(j["module"], ROOT),
                                               "what": "youtokentome.BPE(model, n_threads=8).encode(list[str], OutputType.ID) -> list[list[int]], median of 3"
                                                       + (", taskset -c 0-7" if (pre and which == "reference") else "")}
            except Exception as e:  # noqa: BLE001
                res["encode_list_" + which] = {"error": str(e), "stderr": (r.stderr[-300:] if "r" in dir() else "")}
        a, b = res.get("encode_list_drop_in", {}), res.get("encode_list_reference", {})
        if a.get("value") and b.get("value"):
            res["encode_list_drop_in_over_reference"] = round(a["value"] / b["value"], 2)
            res["encode_list_ids_equal"] = a["ids"] == b["ids"]
        os.remove(lines)
    if host is not None and ctx["args"].size_mb == 1000 and ctx["args"].corpus == "abcd":
        # Whole-process training time, what the reference publishes (tests/speed_test/speed_test.py:71-83, benchmark.md): `yttm bpe` from
        # process start to exit -- interpreter, imports, HIP initialisation, the cold first call and the model file included -- on the 1 GB
        # corpus of configs[1] (in the page cache), the reference's own CLI beside it (n_threads default: 8 for training, bpe.cpp:1348).
        cin = os.path.join(tmpdir, "cli_train.txt")
        with open(cin, "wb") as f:
            f.write(host)
        cli = {}
        for which in ("drop_in", "reference"):
            cmodel = os.path.join(tmpdir, "cli_train_%s.model" % which)
            code = "import sys; from youtokentome.yttm_cli import main; sys.argv = ['yttm', 'bpe', '--data', %r, '--model', %r, '--vocab_size', '%d']; main()" % (cin, cmodel, ctx["args"].vocab)
            try:
                t0 = time.perf_counter()
                r = subprocess.run((pre if which == "reference" else []) + [sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env_for(which))
                dt = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError(r.stderr.decode()[-300:])
                cli[which] = {"seconds": round(dt, 3), "MBps": round(len(host) / 1e6 / dt, 1),
                              "what": "`yttm bpe --data <1 GB file> --model m --vocab_size %d`, process start to exit%s" % (ctx["args"].vocab, ", taskset -c 0-7" if (pre and which == "reference") else "")}
                if which == "drop_in" and ctx["pins"].get("c2_1gb"):
                    cli[which]["model_matches_reference"] = md5_file(cmodel) == ctx["pins"]["c2_1gb"]["model_md5"]
                    out["parity"]["cli_train_model_matches_reference"] = cli[which]["model_matches_reference"]
            except Exception as e:  # noqa: BLE001
                cli[which] = {"error": str(e)}
        if cli.get("drop_in", {}).get("seconds") and cli.get("reference", {}).get("seconds"):
            cli["drop_in_over_reference"] = round(cli["reference"]["seconds"] / cli["drop_in"]["seconds"], 1)
        out.setdefault("e2e", {})["cli_train"] = cli
        os.remove(cin)
    if zhost is not None and out.get("extra", {}).get("zipf"):
        # N1: `yttm encode` on the Zipf corpus' lines, stdin -> stdout (a file; /dev/null would hide the writer)
        zin = os.path.join(tmpdir, "cli_in.txt")
        with open(zin, "wb") as f:
            f.write(zhost)
        zmodel = os.path.join(tempfile.gettempdir(), "yttm_bench_zipf_%s.model" % os.environ.get("MASTER_PORT", str(os.getpid())))
        n_lines = zhost.count(b"\n")
        sums = {}
        for which in ("reference", "drop_in"):
            zout = os.path.join(tmpdir, "cli_out_%s.txt" % which)
            code = "import sys; from youtokentome.yttm_cli import main; sys.argv = ['yttm', 'encode', '--model', %r, '--output_type', 'id', '--n_threads', '8']; main()" % zmodel
            try:
                t0 = time.perf_counter()
                with open(zin, "rb") as fi, open(zout, "wb") as fo:
                    r = subprocess.run((pre if which == "reference" else []) + [sys.executable, "-c", code], stdin=fi, stdout=fo, stderr=subprocess.PIPE, env=env_for(which))
                dt = time.perf_counter() - t0
                if r.returncode != 0:
                    raise RuntimeError(r.stderr.decode()[-300:])
                sums[which] = md5_file(zout)
                res["cli_encode_" + which] = {"value": round(n_lines / dt, 1), "unit": "sentences/s", "MBps_in": round(len(zhost) / 1e6 / dt, 1), "seconds": round(dt, 3),
                                              "output_bytes": os.path.getsize(zout),
                                              "what": "`yttm encode --output_type id --n_threads 8` < the %d lines of the Zipf corpus > file; process start to exit (model load%s included)"
                                                      % (n_lines, ", GPU context" if which == "drop_in" else "")}
                os.remove(zout)
            except Exception as e:  # noqa: BLE001
                res["cli_encode_" + which] = {"error": str(e)}
        if len(sums) == 2:
            res["cli_outputs_identical"] = sums["reference"] == sums["drop_in"]
            out["parity"]["cli_encode_output_identical_to_reference"] = res["cli_outputs_identical"]
        a, b = res.get("cli_encode_drop_in", {}), res.get("cli_encode_reference", {})
        if a.get("value") and b.get("value"):
            res["cli_encode_drop_in_over_reference"] = round(a["value"] / b["value"], 2)
        os.remove(zin)
    return res


def _init_rccl(L, dist, torch, dev, rank, world, local_rank):
    """RCCL communicator for the library: rank 0 creates the unique id, torch.distributed broadcasts it."""
    idbuf = (C.c_uint8 * 128)()
    if rank == 0:
        if L.yttm_comm_rccl_unique_id(idbuf) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
    if dist is not None:
        t = torch.tensor(list(idbuf), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        idbuf = (C.c_uint8 * 128)(*t.cpu().tolist())
    h = C.c_void_p()
    if L.yttm_comm_rccl_create(idbuf, rank, world, local_rank, C.byref(h)) != 0:
        raise RuntimeError("ncclCommInitRank failed")
    return h


if __name__ == "__main__":
    main()
