"""one-off: FastBPE.train to EXHAUSTION (max_vocab beyond reach) on pan_tadeusz and a slice of train-5K, every merge against the oracle"""
import json, sys, time
sys.path.insert(0, ".")
import numpy as np
from subword_tokenizers_amd import _native as N, tokenizers
from oracle import oracle as O
N.init(0)
for name, path, n in (("pan_tadeusz", "tests/golden/ref/data/pan_tadeusz.json", None), ("train-5K[:1500]", "tests/golden/ref/data/train-5K.json", 1500)):
    try:
        sents = json.load(open(path, encoding="utf-8"))
    except FileNotFoundError:
        print(name, "corpus file not here", path); continue
    if n: sents = sents[:n]
    tok = tokenizers.FastBPE(); t0 = time.time(); tok.train(sents, 10 ** 7); t1 = time.time()
    orc = O.OracleBPETrainer(sents); orc.run(10 ** 7); t2 = time.time()
    want = [tuple(m) for m in orc.merges_list]; got = [tuple(m) for m in tok.merges_list]
    bad = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)
    gs, go, gf = tok._trainer.export(); ws, wo, wf = orc.export()
    print(name, "merges", len(got), len(want), "first difference", bad, "stream equal", bool(np.array_equal(gs, ws) and np.array_equal(go, wo)),
          "device %.2f s oracle %.1f s" % (t1 - t0, t2 - t1), tok._trainer.stats())
    assert bad is None and len(got) == len(want) and np.array_equal(gs, ws)
print("EXHAUSTION PARITY OK")
