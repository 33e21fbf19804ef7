"""how many consecutive merges of a training run could share one launch pair ("plateau batching"): an upper and a lower estimate
from the merge sequence alone.  A batch = consecutive merges at one count level whose symbols are pairwise disjoint; the lower
estimate also ends a batch at a member that has a same-level pair (later in the run) touching its outer sides."""
import sys
sys.path.insert(0, ".")
import numpy as np
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
sents = synth.s85k() if (len(sys.argv) < 2 or sys.argv[1] == "lex") else synth.s85k_open()
tok = tokenizers.FastBPE(); tok.train(sents, 8000)
tr = tok._trainer.step_trace()
merges = [tuple(m) for m in tok.merges_list] if hasattr(tok, "merges_list") else [tuple(m) for m in tok.merges]
cnt = tr[:, 0].tolist()
n = len(merges)
for mode in ("upper", "lower"):
    i = 0; batches = []
    while i < n:
        used = set(merges[i]); k = 1
        # same-level run ahead (the tied set is at least these, minus pairs made of symbols born later)
        j_end = i
        while j_end + 1 < n and cnt[j_end + 1] == cnt[i]: j_end += 1
        def dangerous(m, upto):
            a, b = merges[m]
            for q in range(i, upto + 1):
                if q == m: continue
                x, y = merges[q]
                if y == a or x == b: return True
            return False
        while i + k <= j_end and k < 16:
            if mode == "lower" and dangerous(i + k - 1, j_end): break
            nx = merges[i + k]
            made = {merges[q][0] + merges[q][1] for q in range(i, i + k)}
            if nx[0] in used or nx[1] in used or nx[0] in made or nx[1] in made: break
            used.update(nx); k += 1
        batches.append(k); i += k
    b = np.array(batches)
    print(mode, "batches", len(b), "of", n, "merges: mean size %.2f" % b.mean(), "hist", np.bincount(b)[:17].tolist())
    # by training phase
    pos = np.cumsum(b)
    for lo, hi in ((0, 1000), (1000, 3000), (3000, 6000), (6000, n)):
        sel = (pos > lo) & (pos <= hi)
        if sel.any(): print("   merges %d..%d: mean batch %.2f" % (lo, hi, b[sel].mean()))
