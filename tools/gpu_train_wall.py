#!/usr/bin/env python3
"""Where the wall time of FastBPE.train on S85k-open goes on the host side (diagnostics): sections + cProfile."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
corpus = synth.s85k_open()
for rep in range(3):
    t = tokenizers.FastBPE()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t.train(corpus, 8000)
    torch.cuda.synchronize(); print("train wall %.1f ms, merges %d" % ((time.perf_counter() - t0) * 1e3, len(t.merges_list)), flush=True)
# sections
t0 = time.perf_counter(); tr = N.BpeTrainer.from_texts(corpus); torch.cuda.synchronize(); t1 = time.perf_counter()
print("from_texts %.1f ms" % ((t1 - t0) * 1e3))
base = tr.base_symbols(); t2 = time.perf_counter(); print("base_symbols %.2f ms" % ((t2 - t1) * 1e3))
l, r, c = tr.run(8000 - len(base), N.SYM_BASE); t3 = time.perf_counter(); print("run %.1f ms (%d merges)" % ((t3 - t2) * 1e3, len(l)))
print({k: v for k, v in tr.stats().items()} if hasattr(tr, "stats") else "")
tr.close()
t0 = time.perf_counter(); joined = "\x00".join(corpus); t1 = time.perf_counter(); b = joined.encode("utf-8", "surrogatepass"); t2 = time.perf_counter()
print("join %.1f ms, encode %.1f ms, %d bytes" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, len(b)))
pr = cProfile.Profile(); t = tokenizers.FastBPE(); pr.enable(); t.train(corpus, 8000); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
