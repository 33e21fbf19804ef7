#!/usr/bin/env python3
"""Per-call cost of the reference-style entry points (one sentence per call): FastBPE.tokenize / FastWP.tokenize (diagnostics)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
sents = synth.s85k()[:3000]
nb = sum(len(s.encode()) for s in sents)
bpe = tokenizers.FastBPE(); bpe.merges_list = list(synth.pretrained_merges()[:8000]); bpe._build_table()
wp = tokenizers.FastWP(); wp.vocab = set(synth.v30k()); wp._build_trie()
for name, fn in (("FastBPE.tokenize", bpe.tokenize), ("FastWP.tokenize", wp.tokenize), ("FastBPE.encode_ids_batch([s])", lambda s: bpe.encode_ids_batch([s]))):
    for s in sents[:50]:
        fn(s)
    t = time.perf_counter()
    for s in sents:
        try:
            fn(s)
        except RuntimeError:
            pass
    dt = time.perf_counter() - t
    print("%-32s %7.1f us/call  %7.2f MB/s  %8.0f sentences/s" % (name, dt / len(sents) * 1e6, nb / dt / 1e6, len(sents) / dt), flush=True)
