"""`import youtokentome as yttm` on a machine with an MI355X: the reference's public names (youtokentome/__init__.py:1)."""
from .youtokentome import BPE, OutputType  # noqa: F401

__all__ = ["BPE", "OutputType"]
