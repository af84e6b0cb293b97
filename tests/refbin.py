"""Runner for oracle/_ref/yttm_ref_{det,prod}: the UNMODIFIED reference sources compiled by oracle/Makefile.
TEST INFRASTRUCTURE.  The binaries travel to the GPU box (git-ignored, not gpurun-ignored)."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")


def path(kind="det"):
    return os.path.join(REF_DIR, f"yttm_ref_{kind}")


def available(kind="det"):
    if not os.path.exists(path(kind)) and os.path.isdir("/root/reference/youtokentome/cpp"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=False, capture_output=True)
    return os.path.exists(path(kind))


def train(corpus, model, vocab, coverage=1.0, n_threads=1, pad=0, unk=1, bos=2, eos=3, kind="det"):
    r = subprocess.run([path(kind), "train", corpus, model, str(vocab), repr(float(coverage)), str(n_threads),
                        str(pad), str(unk), str(bos), str(eos)], capture_output=True, text=True)
    out = json.loads(r.stdout.strip().splitlines()[-1]) if r.stdout.strip() else {"ok": False, "message": r.stderr}
    if not out.get("ok"):
        raise ValueError(out.get("message", "reference failed"))
    return out


def encode(model, lines_file, n_threads=1, bos=False, eos=False, reverse=False, dropout=0.0, subword=False, kind="prod"):
    args = [path(kind), "encode", model, lines_file, "-", str(n_threads), str(int(bos)), str(int(eos)),
            str(int(reverse)), repr(float(dropout))]
    if subword:
        args.append("subword")
    r = subprocess.run(args, capture_output=True)
    if r.returncode != 0:
        msg = r.stdout.decode(errors="replace")
        try:
            msg = json.loads(msg.strip().splitlines()[-1])["message"]
        except Exception:
            pass
        raise ValueError(msg)
    lines = r.stdout.decode().split("\n")[:-1]
    if subword:
        return [ln.split(" ")[:-1] for ln in lines]
    return [[int(t) for t in ln.split()] for ln in lines]


def encode_bench(model, lines_file, n_threads=8, dropout=0.0, max_lines=-1, kind="prod"):
    r = subprocess.run([path(kind), "encode_bench", model, lines_file, str(n_threads), repr(float(dropout)),
                        str(max_lines)], capture_output=True, text=True)
    return json.loads(r.stdout.strip().splitlines()[-1])
