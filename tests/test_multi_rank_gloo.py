"""N>1 path on CPU: two ranks (torch.distributed/gloo on 127.0.0.1), each with its own corpus shard, exchange char
histograms and pair-count deltas through the library's communicator interface; the model must be byte-identical to the
oracle trained on the whole corpus.  (Kernels run under the HIP emulator; the RCCL transport itself needs GPUs.)"""
import filecmp
import os
import random
import socket
import subprocess
import sys

import pytest

import gen
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(corpus, model, vocab, coverage, world, lib, extra_env=None, sharded=True):
    """sharded=True: the per-round delta exchange (YTTM_REPLICATE_MAX_TOKENS=0 -- these corpora are far below the size from which the
    library shards the merge loop by itself); False: the library's choice for small word tables, the replicated merge loop."""
    port = str(free_port())
    env = dict(os.environ, YTTM_AMD_LIB=lib)
    if sharded:
        env["YTTM_REPLICATE_MAX_TOKENS"] = "0"
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_train_worker.py"), str(r), str(world), port, corpus, model, str(vocab),
                               repr(coverage)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (o, e[-3000:])


@pytest.mark.parametrize("world", [2, 3])
def test_two_ranks_equal_single_oracle(tmp_path, sim_lib, world):
    rng = random.Random(world)
    cases = [(gen.readme_corpus(120, 80, seed=2), 300, 1.0),
             (gen.unicode_text(rng, 4000, "mix", p_invalid=0.01), 90, 0.9),
             (("aaaa aaaaa abababab aabbaabb bbbbbb ab aaab baaa " * 12).encode(), 40, 1.0)]
    for i, (text, vocab, cov) in enumerate(cases):
        corpus = str(tmp_path / f"c{i}.txt")
        open(corpus, "wb").write(text)
        m_mp, m_ora = str(tmp_path / f"mp{i}.model"), str(tmp_path / f"ora{i}.model")
        run_world(corpus, m_mp, vocab, cov, world, sim_lib)
        O.train(text, m_ora, vocab, cov)
        assert filecmp.cmp(m_mp, m_ora, shallow=False), f"case {i}"


@pytest.mark.parametrize("world", [4, 8])
def test_node_sized_worlds_equal_single_oracle(tmp_path, sim_lib, world):
    """The target machine is EIGHT GPUs (BASELINE.json north_star: 1/2/4/8); rounds 1 - 5 only ever ran worlds of 1 .. 3.  Four and
    eight ranks here, one emulator process each: the sharded loop (per-round all-gather of delta blocks; the records after K3 through
    comm_plan.h's offsets, seven peers' runs back to back), the replicated loop, word mode forced on, and a corpus so small that some ranks'
    byte ranges hold no word at all (a rank without words still takes part in every collective): the oracle's model on the whole corpus."""
    rng = random.Random(800 + world)
    small = {"YTTM_WORDS_GRID": "2", "YTTM_WGATHER_GRID": "2", "YTTM_WORD_HINT_FLOOR": "256"}  # (the emulator's time goes with the workgroups; 8 processes share 8 cores)
    force = dict(small, YTTM_WORD_MIN_TILES="0", YTTM_WORD_MIN_TOKENS="0", YTTM_WORD_DIV="0")
    cases = [(gen.readme_corpus(160, 80, seed=12), 260, 1.0, {}, True, ""),
             (gen.unicode_text(rng, 5000, "mix", p_invalid=0.01), 90, 0.9, {}, True, ""),
             (gen.readme_corpus(160, 80, "abcde ", seed=13), 220, 1.0, force, True, "word_rounds>0"),
             (gen.zipf_corpus(30000, vocab=500, seed=9), 260, 1.0, small, False, "replicated_merge_loop==1"),
             (b"ab abab  \n\n   ba ab\n", 12, 1.0, {}, True, "")]  # (23 bytes over 4 / 8 ranks: most ranges are white space or empty)
    for i, (text, vocab, cov, env, sharded, expect) in enumerate(cases):
        corpus = str(tmp_path / f"c{i}.txt")
        open(corpus, "wb").write(text)
        m_mp, m_ora = str(tmp_path / f"mp{i}.model"), str(tmp_path / f"ora{i}.model")
        # (a rank without class-A words follows the word-mode decision without switching: the report check is rank 0's, which always has words here)
        run_world(corpus, m_mp, vocab, cov, world, sim_lib, dict(env, YTTM_TEST_EXPECT_RANK0=expect), sharded=sharded)
        O.train(text, m_ora, vocab, cov)
        assert filecmp.cmp(m_mp, m_ora, shallow=False), f"world {world} case {i}"


@pytest.mark.parametrize("world", [2, 3])
def test_front_end_under_the_upload_on_every_rank(tmp_path, sim_lib, world):
    """Round 5: K1, K2a and the dedup run under the upload on EVERY rank of a multi-GPU run (each on its own byte range; the char histogram's
    all-reduce and the pair exchange come behind, as before) -- forced onto toy shards with parts and I/O chunks of a few KB.  With coverage 1
    every rank takes the word table it made that way (the report says so); with coverage < 1 the common alphabet drops chars and every rank
    redoes its dedup by ids -- same model either way, sharded and replicated."""
    rng = random.Random(40 + world)
    env = {"YTTM_FE_OVERLAP_MIN": "0", "YTTM_FE_PART_KB": "4", "YTTM_IO_CHUNK_KB": "4"}
    cases = [(gen.readme_corpus(400, 90, "abcdef ", seed=8), 260, 1.0, "front_end_overlapped==1"),
             (gen.unicode_text(rng, 9000, "mix", p_invalid=0.01), 120, 1.0, "front_end_overlapped==1"),
             (gen.unicode_text(rng, 9000, "mix", p_invalid=0.01), 100, 0.85, "front_end_overlapped==0")]
    for i, (text, vocab, cov, expect) in enumerate(cases):
        corpus = str(tmp_path / f"c{i}.txt")
        open(corpus, "wb").write(text)
        m_ora = str(tmp_path / f"ora{i}.model")
        O.train(text, m_ora, vocab, cov)
        for sharded in (True, False):
            m_mp = str(tmp_path / f"mp{i}_{int(sharded)}.model")
            run_world(corpus, m_mp, vocab, cov, world, sim_lib, dict(env, YTTM_TEST_EXPECT=expect if sharded else ""), sharded=sharded)
            assert filecmp.cmp(m_mp, m_ora, shallow=False), f"case {i} sharded={sharded}"


@pytest.mark.parametrize("world", [2, 3])
def test_chunked_front_end_on_every_rank(tmp_path, sim_lib, world):
    """A shard that does not fit a rank's HBM is taken in chunks there (gpu_ctx.cpp front_end_chunked); such a rank no longer holds its text,
    so the replicated loop -- which gathers the shards -- is declined by the common verdict and every rank stays sharded: same model."""
    rng = random.Random(70 + world)
    env = {"YTTM_FE_CHUNK_KB": "2"}
    for i, (text, vocab, cov) in enumerate([(gen.readme_corpus(300, 90, "abcdef ", seed=8), 260, 1.0), (gen.unicode_text(rng, 9000, "mix", p_invalid=0.01), 100, 0.85)]):
        corpus, m_ora = str(tmp_path / f"c{i}.txt"), str(tmp_path / f"ora{i}.model")
        open(corpus, "wb").write(text)
        O.train(text, m_ora, vocab, cov)
        for sharded in (True, False):
            m_mp = str(tmp_path / f"mp{i}_{int(sharded)}.model")
            run_world(corpus, m_mp, vocab, cov, world, sim_lib, dict(env, YTTM_TEST_EXPECT="front_end_chunks>1,replicated_merge_loop==0"), sharded=sharded)
            assert filecmp.cmp(m_mp, m_ora, shallow=False), f"case {i} sharded={sharded}"


@pytest.mark.parametrize("world", [1, 2, 3])
def test_replicated_merge_loop_equals_single_oracle(tmp_path, sim_lib, world):
    """Small word tables (the library's own choice below 2^26 dedup tokens): the ranks dedup their shards, gather the shards into the
    whole corpus on every rank and run the merge loop alone -- no collective per round (bpe.cpp:1029-1044: merge the shards' maps once,
    then loop); rank 0 writes the model, byte-identical to the oracle's on the whole corpus."""
    rng = random.Random(40 + world)
    cases = [(gen.readme_corpus(120, 80, seed=3), 300, 1.0),
             (gen.unicode_text(rng, 4000, "mix", p_invalid=0.01), 90, 0.9),
             (gen.zipf_corpus(60000, vocab=900, seed=5), 400, 1.0),
             (("aaaa aaaaa abababab aabbaabb bbbbbb ab aaab baaa " * 12).encode(), 40, 1.0)]
    for i, (text, vocab, cov) in enumerate(cases):
        corpus = str(tmp_path / f"c{i}.txt")
        open(corpus, "wb").write(text)
        m_mp, m_ora = str(tmp_path / f"mp{i}.model"), str(tmp_path / f"ora{i}.model")
        run_world(corpus, m_mp, vocab, cov, world, sim_lib, sharded=False)
        O.train(text, m_ora, vocab, cov)
        assert filecmp.cmp(m_mp, m_ora, shallow=False), f"case {i}"


def test_replication_needs_memory_on_every_rank(tmp_path, sim_lib):
    """the replicated loop gathers the WHOLE corpus on every rank.  Few dedup tokens alone must not choose it: the
    gathered text and the second dedup have to fit beside the shard on every rank, and the verdict is collective -- one rank short of
    memory (YTTM_TEST_FREE_BYTES on rank 1 only) keeps EVERY rank in the sharded loop (nobody waits in a gather the others skipped)."""
    text = gen.zipf_corpus(60000, vocab=900, seed=5)
    corpus = str(tmp_path / "c.txt")
    open(corpus, "wb").write(text)
    m_mp, m_ora = str(tmp_path / "mp.model"), str(tmp_path / "ora.model")
    run_world(corpus, m_mp, 400, 1.0, 3, sim_lib, {"YTTM_TEST_EXPECT": "replicated_merge_loop==1"}, sharded=False)
    run_world(corpus, m_mp, 400, 1.0, 3, sim_lib, {"YTTM_TEST_EXPECT": "replicated_merge_loop==0", "YTTM_TEST_FREE_BYTES_RANK1": "1000"}, sharded=False)
    O.train(text, m_ora, 400, 1.0)
    assert filecmp.cmp(m_mp, m_ora, shallow=False)


@pytest.mark.parametrize("world", [2, 3])
def test_word_mode_on_every_rank(tmp_path, sim_lib, world):
    """K4's word mode changes which words a rank's apply pass visits, not what the ranks exchange: forced on from the second
    round on every rank of the sharded merge loop, one launch per round there too (k_words<FUSED>; and, second pass, with the record
    regions and the record log overflowing), then on every rank of the replicated loop: same model as the oracle on the whole corpus."""
    rng = random.Random(70 + world)
    cases = [(gen.readme_corpus(200, 90, seed=4), 400, 1.0),
             (gen.zipf_corpus(40000, vocab=700, seed=6), 300, 1.0),
             (gen.unicode_text(rng, 6000, "mix", p_invalid=0.01), 90, 0.9)]
    force = {"YTTM_WORD_MIN_TILES": "0", "YTTM_WORD_MIN_TOKENS": "0", "YTTM_WORD_DIV": "0"}
    small = {"YTTM_WORDS_GRID": "3", "YTTM_WGATHER_GRID": "2"}  # (the emulator's time goes with the workgroups)
    for extra, sharded in (({}, True), (dict(small, YTTM_WORD_DREC="16", YTTM_WORD_LOG="200", YTTM_HOT_TARGET="16", YTTM_HOT_MIN="4", YTTM_HOT_CAP="64"), True),
                           (small, False), (dict(small, YTTM_WORD_DREC="16", YTTM_WORD_LOG="200"), False)):
        for i, (text, vocab, cov) in enumerate(cases):
            corpus = str(tmp_path / f"c{i}.txt")
            open(corpus, "wb").write(text)
            m_mp, m_ora = str(tmp_path / f"mp{i}.model"), str(tmp_path / f"ora{i}.model")
            expect = "word_rounds>0,word_fused_rounds>0"  # (every rank's own report: one-launch rounds in the sharded loop too)
            run_world(corpus, m_mp, vocab, cov, world, sim_lib, dict(force, YTTM_TEST_EXPECT=expect, **extra), sharded=sharded)
            O.train(text, m_ora, vocab, cov)
            assert filecmp.cmp(m_mp, m_ora, shallow=False), (i, extra)


@pytest.mark.parametrize("world", [2, 3])
def test_word_mode_switch_is_one_decision(tmp_path, sim_lib, world):
    """the switch to word mode -- and with it the hot list's target, so the thresholds of every later candidate scan,
    and the batch split -- must not be taken from rank-local numbers.  Shards of very different shape: the first ranks hold many distinct
    words (many class-A tiles, many merge sites), the last one a handful of words repeated (one tile) or nothing but white space.  With
    thresholds between the two (YTTM_WORD_MIN_TILES) a rank-local rule would switch the big ranks and not the small one; the decision is
    taken from the sums over the ranks' block headers instead, the same on every rank in the same round: every rank with words reports
    word-mode rounds, the model is the oracle's."""
    rng = random.Random(300 + world)
    big = b" ".join("".join(rng.choice("abcdefgh") for _ in range(rng.randint(2, 14))).encode() for _ in range(9000)) + b"\n"
    few = (b"abab cdcd abcd " * (len(big) // (15 * (world - 1)))) + b"\n"
    blank = b" \n" * (len(big) // (2 * (world - 1)))
    hooks = {"YTTM_WORD_MIN_TILES": "6", "YTTM_WORD_MIN_TOKENS": "0", "YTTM_WORD_DIV": "0", "YTTM_HOT_TARGET": "24", "YTTM_HOT_MIN": "6",
             "YTTM_HOT_TARGET_WORDS": "96", "YTTM_WORDS_GRID": "3", "YTTM_WGATHER_GRID": "2"}
    for i, (tail, extra, expect) in enumerate(((few, {}, "word_rounds>0"), (blank, {}, ""))):
        text = big * (world - 1) + tail  # rank world-1 gets (nearly) only the tail
        corpus = str(tmp_path / f"c{i}.txt")
        open(corpus, "wb").write(text)
        m_mp, m_ora = str(tmp_path / f"mp{i}.model"), str(tmp_path / f"ora{i}.model")
        run_world(corpus, m_mp, 260, 1.0, world, sim_lib, dict(hooks, YTTM_TEST_EXPECT=expect, **extra))
        O.train(text, m_ora, 260, 1.0)
        assert filecmp.cmp(m_mp, m_ora, shallow=False), i


def test_two_ranks_long_words_and_small_hot_list(tmp_path, sim_lib):
    """Words beyond the LDS tile kernels (class C, k_giant.hip) on both ranks, and a hot list so small that it overflows and
    is rebuilt all the time: every rank must reach the same verdict on every overflow (the lists hold the same pairs on every rank:
    k_fold_list) to stay in lock step."""
    rng = random.Random(9)
    long_words = ["".join(rng.choice("abc") for _ in range(n)) for n in (2100, 2600, 3001)]
    lines = []
    for i in range(40):
        lines.append(" ".join(["".join(rng.choice("abcd") for _ in range(rng.randint(1, 9))) for _ in range(12)] + [long_words[i % 3]]))
    text = ("\n".join(lines) + "\n").encode()
    corpus = str(tmp_path / "c.txt")
    open(corpus, "wb").write(text)
    m_mp, m_ora = str(tmp_path / "mp.model"), str(tmp_path / "ora.model")
    run_world(corpus, m_mp, 150, 1.0, 2, sim_lib, {"YTTM_HOT_TARGET": "8", "YTTM_HOT_MIN": "3", "YTTM_HOT_CAP": "32"})
    O.train(text, m_ora, 150, 1.0)
    assert filecmp.cmp(m_mp, m_ora, shallow=False)


def test_exchange_blocks_too_small_are_repeated(tmp_path, sim_lib):
    """Per round the ranks all-gather fixed-size blocks of delta records and fold them in without a host round trip; a rank
    whose records did not fit is skipped by everybody and the exchange is repeated with larger blocks (gpu_ctx.cpp
    settle_exchange).  Blocks of one record force that path in nearly every round -- with and without the hot list."""
    rng = random.Random(21)
    text = gen.unicode_text(rng, 5000, "ascii") + gen.readme_corpus(60, 80, seed=5)
    corpus = str(tmp_path / "c.txt")
    open(corpus, "wb").write(text)
    m_ora = str(tmp_path / "ora.model")
    O.train(text, m_ora, 200, 1.0)
    # (blocks are sized from a prediction with a margin of three: a margin of 0.05 and a floor of one record make nearly every round's too small)
    small = {"YTTM_XCHG_BLK_MIN": "2", "YTTM_XCHG_MARGIN": "0.05", "YTTM_TEST_EXPECT": "exchange_retries>20"}
    # YTTM_XCHG_NOTES=2: the adds' notes of the slots that may have crossed a list threshold overflow in nearly every round -- the fold then
    # walks every delta record of every block instead (and must keep doing so across a repeat)
    for i, env in enumerate((small, dict(small, YTTM_HOT_TARGET="8", YTTM_HOT_MIN="3", YTTM_HOT_CAP="32"),
                             dict(small, YTTM_HOT_TARGET="8", YTTM_HOT_MIN="3", YTTM_HOT_CAP="32", YTTM_XCHG_NOTES="2"), {"YTTM_XCHG_NOTES": "2"})):
        m_mp = str(tmp_path / f"mp{i}.model")
        run_world(corpus, m_mp, 200, 1.0, 3, sim_lib, env)
        assert filecmp.cmp(m_mp, m_ora, shallow=False), env


def test_world_of_one_runs_the_exchange_path(tmp_path, sim_lib):
    """A communicator of size 1 still goes through every collective of the N>1 path (what the GPU suite does with RCCL)."""
    text = gen.readme_corpus(100, 80, seed=8)
    corpus = str(tmp_path / "c.txt")
    open(corpus, "wb").write(text)
    m_mp, m_ora = str(tmp_path / "mp.model"), str(tmp_path / "ora.model")
    run_world(corpus, m_mp, 250, 1.0, 1, sim_lib)
    O.train(text, m_ora, 250, 1.0)
    assert filecmp.cmp(m_mp, m_ora, shallow=False)


def test_delta_table_overflow_stops_every_rank(tmp_path, sim_lib):
    """A rank whose per-round delta table (or send block) is too small cannot tell its peers what it changed: the record count in its
    block's header says so, every rank reads it in the same all-gathered blocks and stops with the same error -- nobody is left waiting in a
    collective the others never post (a hang instead of an error)."""
    text = gen.readme_corpus(100, 80, seed=11)
    corpus = str(tmp_path / "c.txt")
    open(corpus, "wb").write(text)
    port = str(free_port())
    env = dict(os.environ, YTTM_AMD_LIB=sim_lib, YTTM_XCHG_TABLE_CAP="8", YTTM_REPLICATE_MAX_TOKENS="0")
    world = 3
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_train_worker.py"), str(r), str(world), port, corpus, str(tmp_path / "m.model"), "250",
                               "1.0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]  # (a hang would end here)
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 3, (p.returncode, o, e[-2000:])
        assert "ERR" in o and "delta exchange" in o, o
