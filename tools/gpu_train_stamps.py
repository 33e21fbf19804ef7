"""where a merge step's time goes ON the device: needs a library built with SWT_EXTRA_FLAGS=-DSWT_STAMPS (see csrc/swt_bpe_train.hip)
usage: SWT_EXTRA_FLAGS=-DSWT_STAMPS python tools/gpu_train_stamps.py [open|lex]"""
import ctypes, sys, time
sys.path.insert(0, ".")
import numpy as np
import importlib
B = importlib.import_module("subword-tokenizers_amd._build")
B.build()
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
L = ctypes.CDLL(B.LIB_PATH)
fn = L.swt_debug_stamps
fn.argtypes = [ctypes.c_int, ctypes.c_void_p]
sents = synth.s85k() if (len(sys.argv) < 2 or sys.argv[1] == "lex") else synth.s85k_open()
tok = tokenizers.FastBPE(); tok.train(sents, 8000)  # warm
assert fn(0, None) == 0
tok = tokenizers.FastBPE()
t0 = time.time(); tok.train(sents, 8000); wall = time.time() - t0
tr = tok._trainer.step_trace()
n = len(tr)
out = np.zeros(4 * 16384 + 80 + 8192 + 6 * 16384, dtype=np.uint64)
assert fn(1, out.ctypes.data) == 0
span = out[:4 * 16384].reshape(4, 16384).astype(np.float64) / 100.0  # us (100 MHz)
ph = out[4 * 16384:4 * 16384 + 48].reshape(3, 16)
why = out[4 * 16384 + 48:4 * 16384 + 80].astype(np.int64)
kstep = out[4 * 16384 + 80:4 * 16384 + 80 + 8192].view(np.uint32)
pmax = out[4 * 16384 + 80 + 8192:].reshape(6, 16384).astype(np.float64) / 100.0
print("merges", n, "wall s", round(wall, 3), "us/merge", round(wall / n * 1e6, 2))
n_steps = int(tok._trainer.stats()["steps"])
steps = np.arange(1, n_steps + 1) & 16383  # step numbers start at 1; a batch's steps after a re-plan request are no-ops
tie = span[1][steps] - span[0][steps]
app = span[3][steps] - span[2][steps]
gap1 = span[2][steps] - span[1][steps]           # tie end -> apply start
gap2 = np.append(span[0][steps[1:]] - span[3][steps[:-1]], 0.0)  # apply end -> next tie start
seen = (span[0][steps] < 1e15) & (span[2][steps] < 1e15)
live = seen & (app > 0.5)  # a no-op apply returns after one load
print("steps", n_steps, "live", int(live.sum()), "no-op", int((seen & ~live).sum()))
print("on-device span  tie: mean %.2f p50 %.2f   apply: mean %.2f p50 %.2f p95 %.2f   sum/merge %.2f" % (
    tie[live].mean(), np.median(tie[live]), app[live].mean(), np.median(app[live]), np.percentile(app[live], 95), (tie[live].sum() + app[live].sum()) / n))
print("no-op steps: tie %.2f apply %.2f" % (tie[seen & ~live].mean(), app[seen & ~live].mean()))
g2 = gap2[(gap2 > 0) & (gap2 < 100)]
print("boundary  tie->apply: mean %.2f p50 %.2f   apply->tie: mean %.2f p50 %.2f" % (gap1[seen].mean(), np.median(gap1[seen]), g2.mean(), np.median(g2)))
for lo in (0, 1000, 3000, 5000, 7000, 9000):
    print("  step", lo, "tie", tie[lo:lo + 8].round(1), "apply", app[lo:lo + 8].round(1))
k = ph[0][15]
if k:
    print("tie launch, workgroup 0, tied steps (%d): state %.2f  mirror+prefetch %.2f  argmax+set %.2f  first word %.2f  later trips %.2f us;  trips %.2f" % ((k,) + tuple(ph[0][:5] / k / 100.0) + (ph[0][13] / k,)))
    print("   first word = lead stores %.2f + symbols %.2f + probes/hit %.2f" % tuple(ph[0][8:11] / k / 100.0))
k = ph[0][14]
if k:
    print("tie launch, workgroup 0, untied steps (%d): state %.2f  mirror+prefetch %.2f  argmax %.2f us" % ((k,) + tuple(ph[0][5:8] / k / 100.0)))
k = ph[1][15]
if k:
    print("apply launch, first lane done per step after step 512 (%d): prologue+entries %.2f  bounds+claim+stage %.2f  walk %.2f  reserve %.2f  flush %.2f us; deltas %.2f new %.2f" % (
        (k,) + tuple(ph[1][:5] / k / 100.0) + (ph[1][8] / k, ph[1][9] / k)))
k = ph[2][15]
if k:
    print("apply launch, all flushing lanes of steps 1..64 (%d): prologue+entries %.2f  bounds+claim+stage %.2f  walk %.2f  reserve %.2f  flush %.2f us; deltas %.2f new %.2f" % (
        (k,) + tuple(ph[2][:5] / k / 100.0) + (ph[2][8] / k, ph[2][9] / k)))
ns = tr[:, 3].astype(np.int64)
for i in range(0, 12):
    print("  merge %d: count %d, %d merges in unique words, apply span %.1f us -> %.1f merges/us" % (i, tr[i, 0], ns[i] - ns[i + 1], app[i], (ns[i] - ns[i + 1]) / app[i]))
print("tied > 1: %.3f  > 256: %.3f  > 1024: %.3f of the merges; candidates mean %.0f" % (
    (tr[:, 1] > 1).mean(), (tr[:, 1] > 256).mean(), (tr[:, 1] > 1024).mean(), tr[:, 2].mean()))

tot = max(1, int(why[:16].sum()))
print("tied steps by merges carried:", why[1:9].tolist(), " mean %.2f" % (sum(k * int(why[k]) for k in range(16)) / tot))
print("why a batch ended: all seen pairs in %d, beyond the window %d, shared symbol %d, dangerous %d, round-trip cap %d, kMaxBatch %d;  pairs seen per step %.1f of %.1f tied" % (
    tuple(int(x) for x in why[16:22]) + (why[24] / tot, why[25] / tot)))

ks = kstep[steps]
for k in range(0, 9):
    sel = seen & (ks == k) & (np.arange(len(steps)) > 1000)
    if sel.any():
        print("  steps past 1000 that carried %d merges: %5d  tie span %.1f  apply span %.1f us" % (k, int(sel.sum()), tie[sel].mean(), app[sel].mean()))

print("past step 1000: lanes that flushed %d, with more than 8 deltas %d, deltas that did not fit the park (applied one by one) %d" % (int(why[28]), int(why[26]), int(why[27])))

entry = pmax[0][steps] - span[2][steps]   # kernel start -> the last lane to enter apply_body (of the lanes that flushed)
for k in range(1, 9):
    sel = seen & (ks == k) & (np.arange(len(steps)) > 1000) & (pmax[0][steps] > 0)
    if sel.any():
        print("  K=%d: slowest lane per phase: prologue %.1f | entries %.1f  bounds+claim+stage %.1f  walk %.1f  reserve %.1f  flush %.1f us" % (
            (k, entry[sel].mean()) + tuple(pmax[1 + q][steps][sel].mean() for q in range(5))))

# where the device time of the run goes, by training phase (spans + the boundaries after them)
cost = tie + app + gap1 + gap2
cost[~seen] = 0
cum = np.cumsum(ks)  # merges done after each step
print("device time by phase (tie + apply + boundaries), total %.1f ms:" % (cost.sum() / 1e3))
for lo, hi in ((0, 100), (100, 500), (500, 1000), (1000, 2000), (2000, 4000), (4000, 100000)):
    sel = (cum > lo) & (cum <= hi) & seen & (ks > 0)
    if sel.any():
        print("  merges %5d..%5d: %5d steps, %6.1f ms, %.1f us/merge (tie %.1f apply %.1f per step)" % (lo, min(hi, int(cum[-1])), int(sel.sum()), cost[sel].sum() / 1e3, cost[sel].sum() / max(1, ks[sel].sum()), tie[sel].mean(), app[sel].mean()))
noop = seen & (ks == 0)
print("  steps that carried nothing: %d, %.1f ms" % (int(noop.sum()), cost[noop].sum() / 1e3))
