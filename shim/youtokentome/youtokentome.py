"""youtokentome.BPE / youtokentome.OutputType (reference: youtokentome/youtokentome.py:6-99), served by youtokentome_amd."""
from youtokentome_amd.bpe import BPE, OutputType  # noqa: F401
