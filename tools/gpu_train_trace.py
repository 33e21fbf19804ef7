"""per-merge trace of a training run: tie sizes, candidate list length, count levels (diagnostics for DESIGN.md)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
sents = synth.s85k() if (len(sys.argv) < 2 or sys.argv[1] == "lex") else synth.s85k_open()
tok = tokenizers.FastBPE()
t0 = time.time(); tok.train(sents, 8000); print("train s", time.time() - t0)
tr = tok._trainer.step_trace()
print(tok._trainer.stats())
for lo in (0, 100, 500, 1000, 2000, 3000, 5000, 7000, 7900):
    print(lo, "count", tr[lo:lo + 8, 0].tolist(), "tied", tr[lo:lo + 8, 1].tolist(), "ncand", tr[lo:lo + 8, 2].tolist())
print("tied>1 frac", float((tr[:, 1] > 1).mean()), "tied>256 frac", float((tr[:, 1] > 256).mean()), "ncand mean", float(tr[:, 2].mean()), "max", int(tr[:, 2].max()))
