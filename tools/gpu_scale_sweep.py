#!/usr/bin/env python3
"""FastBPE encode throughput against batch size (S85k-shaped text, 8,000 merges): how much of the headline config's time
is per-call overhead (nine launches) rather than per-byte work (diagnostics for DESIGN.md section 5)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
bpe = tokenizers.FastBPE(); bpe.merges_list = list(synth.pretrained_merges()[:8000]); bpe._build_table()
for n_sent in [int(a) for a in sys.argv[1:]] or [10000, 85000, 340000, 850000]:
    sents = synth.sentences(n_sent, 85000)
    text, off = N.pack_utf8([s.lower() for s in sents])
    nb, ns = int(text.size), len(sents)
    d_text = torch.from_numpy(text.copy()).cuda(); d_off = torch.from_numpy(off.view(np.int64).copy()).cuda()
    d_out = torch.empty(nb + 64, dtype=torch.int32, device="cuda"); d_oo = torch.empty(ns + 1, dtype=torch.int64, device="cuda")
    d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
    for flags, name in ((0, "dedup"), (N.BPE_NO_DEDUP, "direct")):
        bpe._table.set_option(N.OPT_DEDUP, N.DEDUP_ALWAYS if name == "dedup" else N.DEDUP_AUTO)  # the dedup pipeline whatever the size, against the direct path
        for _ in range(3):
            bpe._table.encode_dev(d_text.data_ptr(), nb, d_off.data_ptr(), ns, d_out.data_ptr(), d_oo.data_ptr(), d_n.data_ptr(), flags, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            bpe._table.encode_dev(d_text.data_ptr(), nb, d_off.data_ptr(), ns, d_out.data_ptr(), d_oo.data_ptr(), d_n.data_ptr(), flags, 0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print("%8d sentences %7.2f MB  %-6s %9.1f us/call %9.1f MB/s  tokens %d" % (ns, nb / 1e6, name, dt * 1e6, nb / dt / 1e6, int(d_n.item())), flush=True)
    bpe._table.set_option(N.OPT_DEDUP, N.DEDUP_AUTO)
    del d_text, d_off, d_out, d_oo
