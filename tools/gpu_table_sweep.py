#!/usr/bin/env python3
"""Per-call time of the FastWP dedup path (V30k, 1 M sentences) against the size of the word table (SWT_OPT_DEDUP_TABLE_BITS; diagnostics)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
wp = tokenizers.FastWP(); wp.vocab = set(synth.v30k()); wp._build_trie()
n_sent = int(os.environ.get("SWT_SENT", "1000000"))
text, off = synth.wp_corpus(n_sent, seed=1000000, vocab=synth.v30k())
nb = int(text.size)
d_text = torch.from_numpy(text.copy()).cuda(); d_off = torch.from_numpy(off.view(np.int64).copy()).cuda()
d_out = torch.empty(nb + 64, dtype=torch.int32, device="cuda"); d_oo = torch.empty(n_sent + 1, dtype=torch.int64, device="cuda")
d_st = torch.empty(n_sent + 8, dtype=torch.uint8, device="cuda"); d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
def call():
    wp._trie.encode_dev(d_text.data_ptr(), nb, d_off.data_ptr(), n_sent, d_out.data_ptr(), d_oo.data_ptr(), d_st.data_ptr(), d_n.data_ptr(), 0)
ref = None
for bits in [int(a) for a in sys.argv[1:]] or [0, 22, 21, 20, 19, 18, 17]:
    wp._trie.set_option(N.OPT_DEDUP_TABLE_BITS, bits)
    for _ in range(3): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): call()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    ids = d_out[: int(d_n.item())].cpu().numpy().copy()
    if ref is None: ref = ids
    print("table bits %2d  %8.1f us/call  %8.1f MB/s  same=%s" % (bits, dt * 1e6, nb / dt / 1e6, np.array_equal(ref, ids)), flush=True)
wp._trie.set_option(N.OPT_DEDUP_TABLE_BITS, 0)
