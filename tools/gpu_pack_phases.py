"""host-side cost of FastBPE.train's front end on S85k-open: the pieces of pack_and_lower and BpeTrainer.from_text"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from subword_tokenizers_amd import _native as N, synth
import ctypes as C
N.init(0)
texts = synth.s85k_open()
n = len(texts)
def T(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); best = min(best, time.perf_counter() - t0)
    return r, best * 1e3
_, t = T(lambda: np.fromiter(map(len, texts), dtype=np.uint64, count=n)); print("fromiter(map(len)) %.2f ms" % t)
j, t = T(lambda: "".join(texts)); print("join %.2f ms" % t)
d, t = T(lambda: j.encode("utf-8", "surrogatepass")); print("encode %.2f ms (%d bytes)" % (t, len(d)))
b, t = T(lambda: np.frombuffer(d, dtype=np.uint8).copy()); print("frombuffer+copy %.2f ms" % t)
_, t = T(lambda: N.pack_and_lower(texts)); print("pack_and_lower total %.2f ms" % t)
text, off = N.pack_and_lower(texts)
tr, t = T(lambda: N.BpeTrainer.from_text(text, off)); print("from_text %.2f ms" % t)
N.profile_enable(3) if hasattr(N, "profile_enable") else None
