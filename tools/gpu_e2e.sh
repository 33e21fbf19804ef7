# end to end from list[str]: the joined entry points' parity, then the bench line's end_to_end numbers and a phase split
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "joined or edge_shapes or lowercase or ragged or wp_fuzz or pack_and_lower" 2>&1 | tail -5 && \
timeout -k 10 300 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from subword_tokenizers_amd import _native as N, synth, tokenizers
N.init(0)
bpe = tokenizers.FastBPE(); bpe.merges_list = list(synth.pretrained_merges()[:8000]); bpe._build_table()
for name, sents in (("open", synth.s85k_open()), ("lex", synth.s85k())):
    nb = sum(len(s.encode("utf-8", "surrogatepass")) for s in sents)
    for rep in range(4):
        t0 = time.perf_counter(); j, nn = N.join_texts(sents); t1 = time.perf_counter()
        got = bpe._table.encode_joined(j, len(sents)); t2 = time.perf_counter()
        ids, off = bpe.encode_ids_batch(sents); t3 = time.perf_counter()
        print("%s rep %d: join %.2f ms, encode_joined %.2f ms, encode_ids_batch %.2f ms = %.0f MB/s" % (name, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, nb / 1e6 / (t3 - t2)), flush=True)
    t0 = time.perf_counter(); d = "\x00".join(sents).encode("utf-8", "surrogatepass"); t1 = time.perf_counter()
    print("python join+encode %.2f ms" % ((t1 - t0) * 1e3))
PY
