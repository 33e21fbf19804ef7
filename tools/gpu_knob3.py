#!/usr/bin/env python3
"""Whole-step time of the dedup encode path against the tile size of the unique-word pass (SWT_OPT_UNIQUE_TILE; diagnostics)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
bpe = tokenizers.FastBPE(); bpe.merges_list = list(synth.pretrained_merges()[:8000]); bpe._build_table()
sents = synth.s85k()
text, off = N.pack_utf8([s.lower() for s in sents])
nb, ns = int(text.size), len(sents)
d_text = torch.from_numpy(text.copy()).cuda(); d_off = torch.from_numpy(off.view(np.int64).copy()).cuda()
d_out = torch.empty(nb + 64, dtype=torch.int32, device="cuda"); d_oo = torch.empty(ns + 1, dtype=torch.int64, device="cuda")
d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
ref = None
for k in [int(a) for a in sys.argv[1:]] or [0, 256, 64]:
    bpe._table.set_option(N.OPT_UNIQUE_TILE, k)
    for _ in range(5):
        bpe._table.encode_dev(d_text.data_ptr(), nb, d_off.data_ptr(), ns, d_out.data_ptr(), d_oo.data_ptr(), d_n.data_ptr(), 0, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        bpe._table.encode_dev(d_text.data_ptr(), nb, d_off.data_ptr(), ns, d_out.data_ptr(), d_oo.data_ptr(), d_n.data_ptr(), 0, 0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    ids = d_out[: int(d_n.item())].cpu().numpy().copy()
    if ref is None: ref = ids
    print("tile=%4d  %8.1f us/step  %8.1f MB/s  same=%s" % (k, dt * 1e6, nb / dt / 1e6, np.array_equal(ref, ids)), flush=True)
bpe._table.set_option(N.OPT_UNIQUE_TILE, 0)
