#!/usr/bin/env python3
"""Where does FastBPE.train spend its time? (diagnostics)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from subword_tokenizers_amd import _native as N, synth

N.init(0)
sents = synth.s85k()
t0 = time.perf_counter(); low = [s.lower() for s in sents]; t1 = time.perf_counter()
text, off = N.pack_utf8(low); t2 = time.perf_counter()
tr = N.BpeTrainer.from_text(text, off); torch.cuda.synchronize(); t3 = time.perf_counter()
print("lower %.1f ms  pack %.1f ms  create(split+dedup+upload+histogram) %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), tr.info(), flush=True)
base = len(tr.base_symbols())
for chunk in (1, 10, 100, 1000, 3000, 3700):
    t = time.perf_counter()
    l, r, c = tr.run(chunk, 0x110000 + 0)  # ids irrelevant for timing (fresh ids each step)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("run(%4d): %d steps in %.2f ms = %.1f us/step, last count %s, info %s" % (chunk, len(l), dt*1e3, dt*1e6/max(len(l),1), c[-1] if len(c) else None, tr.info()), flush=True)
# stepwise for comparison
tr2 = N.BpeTrainer.from_text(text, off)
t = time.perf_counter()
for i in range(500):
    l, r, c, tied, pos = tr2.best(); tr2.apply(l, r, 0x110000 + i)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("stepwise 500: %.1f us/step" % (dt*1e6/500))
