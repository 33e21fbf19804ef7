#!/usr/bin/env python3
"""FastWP encode: dedup pipeline against the direct path by batch size (V30k vocabulary; sets the dedup threshold; diagnostics)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
wp = tokenizers.FastWP(); wp.vocab = set(synth.v30k()); wp._build_trie()
for n_sent in [int(a) for a in sys.argv[1:]] or [5000, 10000, 20000, 40000, 80000]:
    text, off = synth.wp_corpus(n_sent, seed=1000000, vocab=synth.v30k())
    nb = int(text.size)
    d_text = torch.from_numpy(text.copy()).cuda(); d_off = torch.from_numpy(off.view(np.int64).copy()).cuda()
    d_out = torch.empty(nb + 64, dtype=torch.int32, device="cuda"); d_oo = torch.empty(n_sent + 1, dtype=torch.int64, device="cuda")
    d_st = torch.empty(n_sent + 8, dtype=torch.uint8, device="cuda"); d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
    for knob, name in ((N.DEDUP_ALWAYS, "dedup"), (N.DEDUP_NEVER, "direct")):
        wp._trie.set_option(N.OPT_DEDUP, knob)
        call = lambda: wp._trie.encode_dev(d_text.data_ptr(), nb, d_off.data_ptr(), n_sent, d_out.data_ptr(), d_oo.data_ptr(), d_st.data_ptr(), d_n.data_ptr(), 0)
        for _ in range(3): call()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): call()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("%7d sentences %6.2f MB  %-6s %8.1f us/call %9.1f MB/s" % (n_sent, nb / 1e6, name, dt * 1e6, nb / dt / 1e6), flush=True)
    wp._trie.set_option(N.OPT_DEDUP, N.DEDUP_AUTO)
