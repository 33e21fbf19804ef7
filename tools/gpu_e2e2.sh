# tokenize_batch (strings out): parity of the callers that use it, then its end-to-end rate
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "golden or joined or edge_shapes or train5k or odd_vocab" 2>&1 | tail -5 && \
timeout -k 10 300 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
bpe = tokenizers.FastBPE(); bpe.merges_list = list(synth.pretrained_merges()[:8000]); bpe._build_table()
sents = synth.s85k_open()
nb = sum(len(s.encode("utf-8", "surrogatepass")) for s in sents)
out = None
for rep in range(3):
    out = None
    t0 = time.perf_counter(); out = bpe.tokenize_batch(sents); t1 = time.perf_counter()
    print("FastBPE.tokenize_batch %.1f ms = %.0f MB/s (%d tokens)" % ((t1 - t0) * 1e3, nb / 1e6 / (t1 - t0), sum(map(len, out))), flush=True)
ids, off = bpe.encode_ids_batch(sents)
toks = bpe.decode_ids(ids)
assert out == [toks[int(off[i]):int(off[i + 1])] for i in range(len(sents))]
wp = tokenizers.FastWP(); wp.load_resources(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests/golden/ref/resources/pretrained/FastWordPiece"))
for rep in range(3):
    out = None
    t0 = time.perf_counter(); out = wp.tokenize_batch(sents); t1 = time.perf_counter()
    print("FastWP.tokenize_batch %.1f ms = %.0f MB/s (%d tokens)" % ((t1 - t0) * 1e3, nb / 1e6 / (t1 - t0), sum(map(len, out))), flush=True)
t0 = time.perf_counter(); ids, off, st = wp.encode_ids_batch(sents); t1 = time.perf_counter()
print("FastWP.encode_ids_batch %.1f ms = %.0f MB/s" % ((t1 - t0) * 1e3, nb / 1e6 / (t1 - t0)))
toks = wp._decode(ids)
assert out == [toks[int(off[i]):int(off[i + 1])] for i in range(len(sents))]
PY
