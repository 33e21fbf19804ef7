"""where FastBPE.train's wall time goes on the host side: packing, device census + histogram + index, the merge loop, strings"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
sents = synth.s85k() if (len(sys.argv) < 2 or sys.argv[1] == "lex") else synth.s85k_open()
tok = tokenizers.FastBPE(); tok.train(sents, 8000)  # warm
for rep in range(2):
    t0 = time.perf_counter()
    text, off = N.pack_and_lower(sents)
    t1 = time.perf_counter()
    tr = N.BpeTrainer.from_text(text, off)
    t2 = time.perf_counter()
    base = tr.base_symbols()
    want = 8000 - len(base)
    l, r, c = tr.run(want, N.SYM_BASE)
    t3 = time.perf_counter()
    syms = tokenizers._SymbolTable(); vocab = set(chr(int(x)) for x in base); merges = []; done = set()
    for i in range(len(l)):
        a, b = int(l[i]), int(r[i])
        done.add((a, b))
        ls, rs = syms.string(a), syms.string(b)
        syms.intern(ls + rs); vocab.add(ls + rs); merges.append((ls, rs))
    t4 = time.perf_counter()
    st = tr.stats()
    tr.close()
    t5 = time.perf_counter()
    print("pack+lower %.1f ms | from_text (census, histogram, index) %.1f ms | run %d merges %.1f ms (%.2f us/merge, %d steps, %d replans) | strings %.1f ms | close %.1f ms | total %.1f ms" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(l), (t3 - t2) * 1e3, (t3 - t2) / len(l) * 1e6, st["steps"], st["replans"], (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t5 - t0) * 1e3))
t0 = time.perf_counter(); tok = tokenizers.FastBPE(); tok.train(sents, 8000); print("FastBPE.train %.1f ms" % ((time.perf_counter() - t0) * 1e3))
