"""subword-tokenizers_amd -- the phtryll/subword-tokenizers hot path on AMD MI355X (gfx950).

The directory name carries a hyphen (it mirrors the reference's repository name), so import it through the
alias package `subword_tokenizers_amd` at the repository root.
"""
from ._native import NoDeviceError, SwtError  # noqa: F401
from .tokenizers import FastBPE, FastWP, NaiveBPE, NaiveWP, SubwordTokenizer, TrieView  # noqa: F401

__all__ = ["SubwordTokenizer", "NaiveBPE", "FastBPE", "NaiveWP", "FastWP", "TrieView", "SwtError", "NoDeviceError"]
