#!/usr/bin/env python3
"""Ablation timing of bpe_wordref_kernel (dedup path) on the S85k batch (diagnostics)."""
import os, sys
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
sents = synth.s85k()
text, off = N.pack_utf8([s.lower() for s in sents])
nb, ns = int(text.size), len(sents)
d_text = torch.from_numpy(text.copy()).cuda(); d_off = torch.from_numpy(off.view(np.int64).copy()).cuda()
d_out = torch.empty(nb + 64, dtype=torch.int32, device="cuda"); d_oo = torch.empty(ns + 1, dtype=torch.int64, device="cuda")
d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
def run(knob, reps=4):
    N.check(ABL(2, knob))
    for _ in range(3):
        bpe._table.encode_dev(d_text.data_ptr(), nb, d_off.data_ptr(), ns, d_out.data_ptr(), d_oo.data_ptr(), d_n.data_ptr(), 0, 0)
    torch.cuda.synchronize(); N.profile_enable(3); N.profile_read()
    for _ in range(reps):
        bpe._table.encode_dev(d_text.data_ptr(), nb, d_off.data_ptr(), ns, d_out.data_ptr(), d_oo.data_ptr(), d_n.data_ptr(), 0, 0)
    torch.cuda.synchronize(); ms, n = N.profile_read(); N.profile_enable(False)
    return ms / n * 1e3
for name, k in [("full", 0), ("no word lanes", 1), ("count CAS", 4)]:
    print("%-32s knob=%2d  %8.1f us" % (name, k, run(k)), flush=True)
ABL(2, 0)
