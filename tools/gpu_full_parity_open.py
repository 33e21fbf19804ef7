"""one-off: FastBPE.train on S85k-open to vocab 8,000, every merge against the C oracle (about two minutes of host time)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from subword_tokenizers_amd import _native as N, synth, tokenizers
from oracle import oracle as O
N.init(0)
sents = synth.s85k_open()
tok = tokenizers.FastBPE(); tok.train(sents, 8000)
t0 = time.time()
orc = O.OracleBPETrainer(sents); orc.run(8000)
print("oracle %.1f s" % (time.time() - t0), flush=True)
want = [tuple(m) for m in orc.merges_list]; got = [tuple(m) for m in tok.merges_list]
bad = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)
print("merges", len(got), len(want), "first difference", bad)
gs, go, gf = tok._trainer.export(); ws, wo, wf = orc.export()
print("stream equal", bool(np.array_equal(gs, ws) and np.array_equal(go, wo)), "stats", tok._trainer.stats())
assert bad is None and len(got) == len(want) and np.array_equal(gs, ws)
print("FULL PARITY OK")
