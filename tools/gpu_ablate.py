#!/usr/bin/env python3
"""Ablation timing of bpe_encode_kernel phases on the S85k batch (diagnostics; results are wrong under a knob)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from subword_tokenizers_amd import _native as N, synth, tokenizers

N.init(0)
# the ablation switches exist only in a library built with -DSWT_ABLATION (add it to _build.FLAGS); results are wrong under them
import ctypes
ABL = ctypes.CDLL(N._build.LIB_PATH).swt_ablation_knob
bpe = tokenizers.FastBPE(); bpe.merges_list = list(synth.pretrained_merges()[:8000]); bpe._build_table()
sents = synth.s85k_open() if os.environ.get("SWT_ABLATE_OPEN") else synth.s85k()
text, off = N.pack_utf8([s.lower() for s in sents])
nb, ns = int(text.size), len(sents)
d_text = torch.from_numpy(text.copy()).cuda(); d_off = torch.from_numpy(off.view(np.int64).copy()).cuda()
d_out = torch.empty(nb + 64, dtype=torch.int32, device="cuda"); d_oo = torch.empty(ns + 1, dtype=torch.int64, device="cuda")
d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
print("occupancy (blocks/CU) packed, wide:", N.lib().swt_debug_occupancy(0), N.lib().swt_debug_occupancy(1), flush=True)
def run(knob, reps=20):
    N.check(ABL(0, knob))
    for _ in range(3):
        bpe._table.encode_dev(d_text.data_ptr(), nb, d_off.data_ptr(), ns, d_out.data_ptr(), d_oo.data_ptr(), d_n.data_ptr(), 0, 0)
    torch.cuda.synchronize(); N.profile_enable(True); N.profile_read()
    for _ in range(reps):
        bpe._table.encode_dev(d_text.data_ptr(), nb, d_off.data_ptr(), ns, d_out.data_ptr(), d_oo.data_ptr(), d_n.data_ptr(), 0, 0)
    torch.cuda.synchronize(); ms, n = N.profile_read(); N.profile_enable(False)
    return ms / n * 1e3
bpe._table.set_option(N.OPT_DEDUP, N.DEDUP_AUTO if os.environ.get("SWT_ABLATE_DEDUP") else N.DEDUP_NEVER)  # default: the direct path; SWT_ABLATE_DEDUP=1: the unique-word pass
for name, k in [("full", 0), ("stage only", 1), ("no class table", 2), ("no first-round lookups", 4), ("no merge loop", 8),
                ("no word phase D", 16), ("no compaction/record", 32), ("no D, no lookups", 20), ("no D/lookups/cls", 22),
                ("B only (no C-lookups, D, E)", 52)]:
    print("%-32s knob=%2d  %8.1f us" % (name, k, run(k)), flush=True)
for r in (1, 2, 3, 4, 6, 8, 10, 12, 16, 24):
    print("max rounds %2d   knob=%5d  %8.1f us" % (r, r << 8, run(r << 8)), flush=True)
ABL(0, 0)
