"""Where does the FastBPE device path differ from the oracle?  (diagnosis on the box: first mismatching sentence, both lists)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subword_tokenizers_amd as S
from oracle import oracle as O
from subword_tokenizers_amd import _native as N

N.init(0)
ref = os.path.join(ROOT, "tests", "golden", "ref")
sents = json.load(open(os.path.join(ref, "data", "pan_tadeusz.json"), encoding="utf-8"))
bpe = S.FastBPE()
bpe.load_resources(os.path.join(ref, "resources", "pretrained", "FastBPE"))
orc = O.OracleBPE(bpe.merges_list)
for count in (1, 3, 40, 989):
    ss = sents[:count]
    oids, ooff = orc.tokenize_batch_ids(ss)
    ids, off = bpe.encode_ids_batch(ss)
    ok = np.array_equal(ids, oids) and np.array_equal(off, ooff)
    print("n_sent", count, "ok" if ok else "DIFF", "tokens", ids.size, "oracle", oids.size, flush=True)
    if not ok:
        for s in range(count):
            a = ids[int(off[s]):int(off[s + 1])] if s + 1 < len(off) else ids[0:0]
            b = oids[int(ooff[s]):int(ooff[s + 1])]
            if not np.array_equal(a, b) or off[s] != ooff[s]:
                print(" first differing sentence", s, repr(ss[s])[:80], "off", int(off[s]), int(ooff[s]))
                print("  dev", [hex(int(x)) for x in a[:24]])
                print("  orc", [hex(int(x)) for x in b[:24]])
                break
        print(" off tail dev", off[-3:], "orc", ooff[-3:])
