"""Run by tests/test_sim_asan.py in a subprocess with the AddressSanitizer build of the emulated product sources."""
import pathlib
import random
import sys
import tempfile

sys.path.insert(0, str(pathlib.Path(__file__).parent))
sys.path.insert(0, str(pathlib.Path(__file__).parent.parent))
import gen  # noqa: E402
import stage_checks as S  # noqa: E402

tmp = pathlib.Path(tempfile.mkdtemp())
for name in ("readme_small", "runs", "mix_cov"):
    S.check_golden_train(name, tmp)
for name in S.golden_encode_names()[:3]:
    S.check_golden_encode(name)
rng = random.Random(3)
words = ["".join(rng.choice("abc") for _ in range(n)) for n in (2046, 1500, 1024, 700, 300, 257, 65, 64, 63, 1)]
text = (" ".join(words) + "\n") * 3 + gen.readme_corpus(50, 80).decode()
model = S.check_train_vs_oracle(text.encode(), 200, tmp, tag="lw")
S.check_encode_vs_oracle(model, [" ".join(words), "ab" * 700, "a"], flags=((0, 0, 0), (1, 1, 1)))
S.check_very_long_words(tmp, lengths=(2047, 2048, 2049, 3000))
S.check_encode_mixed_shapes(n_sent=60)
# K4's register paths: a rare pair at many tile positions, many sites per tile, ids behind the LDS flag bitmap
S.check_site_placements(trials=25, seed=8)
many = ["ab" * k for k in range(60, 125, 13)] + ["a" * k for k in range(150, 250, 29)]
S.check_merge_rounds((" ".join(many) + " ").encode(), rounds=5, seed=4)
S.check_merge_rounds(S.texts_small(5, n=1, size=1200)[0], rounds=4, seed=1, id_shift=40000)
# N4: the word cache's table, occurrence array, queues and the end-to-end tokenizer of K5's word mode
S.check_encode_word_cache(n_sent=40)
for t in S.texts_by_alphabet_size(sizes=(5, 33, 64, 70), n_words=200):  # K3's kernels
    S.check_word_table_and_pairs(t)
# K4's word mode (k_words<FUSED>: rule runs, claimed-word list and list allotment in LDS; k_wgather + k_words + k_delta_apply; record regions
# and a record log that overflow; a batch of hundreds of disjoint pairs cut in two)
import os  # noqa: E402
os.environ.update({"YTTM_WORD_MIN_TILES": "0", "YTTM_WORD_MIN_TOKENS": "0", "YTTM_WORD_DIV": "0", "YTTM_WORDS_GRID": "3", "YTTM_WGATHER_GRID": "2"})
wm_text = gen.readme_corpus(200, 90, seed=8)
for cfg in ({}, {"YTTM_WORDS_FUSE_MAX": "0"}, {"YTTM_WORD_LOG": "300", "YTTM_WORD_DREC": "16"}, {"YTTM_WORDS_FUSE_MAX": "0", "YTTM_WORDS_INLINE_MAX": "0", "YTTM_WORD_DREC": "16"}):
    os.environ.update(cfg)
    S.check_train_vs_oracle(wm_text, 500, tmp, tag="wm")
    for k in cfg:
        del os.environ[k]
S.check_train_vs_oracle(gen.disjoint_words_corpus(200), 4 + 800 + 500, tmp, tag="wmsplit")
for k in ("YTTM_WORD_MIN_TILES", "YTTM_WORD_MIN_TOKENS", "YTTM_WORD_DIV", "YTTM_WORDS_GRID", "YTTM_WGATHER_GRID"):
    del os.environ[k]
print("ASAN_SCENARIOS_OK")
