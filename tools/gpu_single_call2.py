#!/usr/bin/env python3
"""Where the time of one reference-style tokenize(text) call goes (diagnostics)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
bpe = tokenizers.FastBPE(); bpe.merges_list = list(synth.pretrained_merges()[:8000]); bpe._build_table()
wp = tokenizers.FastWP(); wp.load_resources(os.path.join(ROOT, "tests/golden/ref/resources/pretrained/FastWordPiece"))
sents = synth.s85k_open()[:2000]
def per_call(f, n=2000):
    for s in sents[:200]: f(s)
    t0 = time.perf_counter()
    for s in sents[:n]: f(s)
    return (time.perf_counter() - t0) / n * 1e6
print("FastBPE.tokenize        %.1f us/call" % per_call(bpe.tokenize))
print("FastWP.tokenize         %.1f us/call" % per_call(wp.tokenize))
print("  encode_ids_batch([s]) %.1f us" % per_call(lambda s: bpe.encode_ids_batch([s])))
print("  pack_and_lower([s])   %.1f us" % per_call(lambda s: N.pack_and_lower([s])))
packed = [N.pack_and_lower([s]) for s in sents]
it = iter(packed * 3)
print("  table.encode          %.1f us" % per_call(lambda s: bpe._table.encode(*next(it))))
ids = [bpe._table.encode(*p)[0] for p in packed]
it2 = iter(ids * 3)
print("  decode_ids            %.1f us" % per_call(lambda s: bpe.decode_ids(next(it2))))
print("mean sentence: %.0f bytes, %.1f tokens" % (np.mean([p[0].size for p in packed]), np.mean([i.size for i in ids])))
