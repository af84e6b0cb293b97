"""youtokentome_amd -- MI355X-native BPE train + encode core, drop-in for `youtokentome` (VKCOM/YouTokenToMe).

    import youtokentome_amd as yttm
    yttm.BPE.train(data="train.txt", model="m.yttm", vocab_size=32000)
    bpe = yttm.BPE("m.yttm"); bpe.encode(["some text"], output_type=yttm.OutputType.ID)

The hot path (char histogram, word dedup, pair count, merge apply, batch encode) runs as hand-written HIP kernels for
gfx950 behind the C ABI of include/yttm_mi355x.h; there is no CPU fallback."""
from .bpe import BPE, OutputType  # noqa: F401

__all__ = ["BPE", "OutputType"]
