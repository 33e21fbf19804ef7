"""how many consecutive UNTIED merges (strictly falling counts) share no symbol: what batching the top-K distinct counts could carry"""
import sys
sys.path.insert(0, ".")
import numpy as np
from subword_tokenizers_amd import _native as N, synth
N.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "open"
if which == "1g":
    sym, off, freq = synth.train_words(2_000_000, 1073741824, total_tokens=2_000_000 * 55)
    tr = N.BpeTrainer.from_words(sym, off, freq); n = 32000
else:
    from subword_tokenizers_amd import tokenizers
    sents = synth.s85k_open() if which == "open" else synth.s85k()
    text, o = N.pack_and_lower(sents); tr = N.BpeTrainer.from_text(text, o); n = 7922
l, r, c = tr.run(n, N.SYM_BASE)
l = l.tolist(); r = r.tolist(); c = c.tolist()
tied = tr.step_trace()[:, 1]
m = len(l)
i = 0; steps = 0; untied_merges = 0; untied_steps = 0; hist = np.zeros(20, dtype=np.int64)
while i < m:
    if tied[i] > 1:
        # a tied step as the device batches it: skip to the next count level change or keep as is (not estimated here)
        j = i + 1
        while j < m and c[j] == c[i] and tied[j] > 1: j += 1
        i = j; continue
    used = {l[i], r[i], N.SYM_BASE + i}; k = 1
    while i + k < m and k < 16 and tied[i + k] <= 1 and c[i + k] < c[i + k - 1]:
        a, b = l[i + k], r[i + k]
        if a in used or b in used: break
        used.update((a, b, N.SYM_BASE + i + k)); k += 1
    hist[k] += 1; untied_merges += k; untied_steps += 1; i += k
print(which, "untied merges", untied_merges, "of", m, "in", untied_steps, "batches: mean %.2f" % (untied_merges / max(1, untied_steps)), "hist", hist[1:12].tolist())
