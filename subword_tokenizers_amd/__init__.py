"""Import alias: the product package lives in `subword-tokenizers_amd/` (hyphenated, not a valid module name).

`import subword_tokenizers_amd` resolves submodules from that directory.
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "subword-tokenizers_amd")]

from ._native import NoDeviceError, SwtError  # noqa: E402,F401
from .tokenizers import FastBPE, FastWP, NaiveBPE, NaiveWP, SubwordTokenizer, TrieView  # noqa: E402,F401

__all__ = ["SubwordTokenizer", "NaiveBPE", "FastBPE", "NaiveWP", "FastWP", "TrieView", "SwtError", "NoDeviceError"]
