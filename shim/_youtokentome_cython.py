"""Drop-in for the reference's extension module `_youtokentome_cython` (youtokentome/cpp/yttm.pyx:51-182): the same class
`BPE` with the same methods (train, encode, decode, subword_to_id, id_to_subword, vocab_size, vocab, encode_cli, decode_cli,
vocab_cli), implemented on the MI355X C ABI (include/yttm_mi355x.h).  The reference's own `youtokentome/youtokentome.py`
and `youtokentome/yttm_cli.py` import this module name and run on it unchanged; so do its tests
(tests/unit_tests/test_python_api.py, test_cli.py) -- see tests/test_reference_suite.py."""
from youtokentome_amd.bpe import _Core as BPE  # noqa: F401
