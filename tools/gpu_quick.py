#!/usr/bin/env python3
"""Quick on-GPU sanity run (development aid): parity of every device path against the CPU oracle + rough timings."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import subword_tokenizers_amd as S  # noqa: E402
from subword_tokenizers_amd import _native as N  # noqa: E402
from oracle import oracle as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def ld(rel):
    with open(os.path.join(G, rel), encoding="utf-8") as f:
        return json.load(f)


def main():
    print("devices", N.device_count(), N.device_info(), flush=True)
    pan = ld("ref/data/pan_tadeusz.json")
    t5k = ld("ref/data/train-5K.json")
    gold = ld("ref/data/pan_tadeusz.tokens.json")
    fails = 0

    # ---- BPE encode
    bpe = S.FastBPE()
    bpe.load_resources(os.path.join(G, "ref/resources/pretrained/FastBPE"))
    t = time.time(); out = bpe.tokenize_batch(pan); dt = time.time() - t
    ok = out == gold["FastBPE"]; fails += not ok
    print("BPE pan_tadeusz vs golden:", ok, "%.3fs" % dt, flush=True)
    orc = O.OracleBPE(bpe.merges_list)
    ids, off = bpe.encode_ids_batch(t5k)
    oids, ooff = orc.tokenize_batch_ids(t5k)
    ok = np.array_equal(ids, oids) and np.array_equal(off, ooff); fails += not ok
    print("BPE train-5K ids vs oracle:", ok, ids.size, oids.size, flush=True)
    if not ok:
        n = min(ids.size, oids.size)
        bad = np.nonzero(ids[:n] != oids[:n])[0]
        print("  first mismatch", bad[:5], ids[bad[:5]], oids[bad[:5]])
        boff = np.nonzero(off != ooff)[0]
        print("  off mismatch", boff[:5])
    fz = ld("fuzz_bpe.json")
    texts = [c["text"] for c in fz["sentences"]]
    out = bpe.tokenize_batch(texts)
    bad = [i for i, c in enumerate(fz["sentences"]) if out[i] != c["pretrained"]]
    fails += bool(bad)
    print("BPE fuzz:", len(texts) - len(bad), "/", len(texts), flush=True)
    for i in bad[:3]:
        print("   ", repr(texts[i][:60]), out[i][:8], fz["sentences"][i]["pretrained"][:8])
    ew = [(c["word"], bpe.encode_word(c["word"]), c["pretrained"]) for c in fz["encode_word"]]
    badw = [x for x in ew if x[1] != x[2]]; fails += bool(badw)
    print("BPE encode_word:", len(ew) - len(badw), "/", len(ew), badw[:2], flush=True)
    # long inputs: long sentence, giant word
    longs = ["słowo " * 3000, "x" * 20000, "ab" * 6000 + " " + "nie " * 10, "", "a", " ".join(pan[:200])]
    ids, off = bpe.encode_ids_batch(longs)
    oids, ooff = orc.tokenize_batch_ids(longs)
    ok = np.array_equal(ids, oids) and np.array_equal(off, ooff); fails += not ok
    print("BPE long/giant:", ok, ids.size, oids.size, flush=True)
    # throughput on a bigger batch (host-buffer path, includes H2D/D2H)
    big = t5k * 20
    text, boff = N.pack_utf8([s.lower() for s in big])
    bpe._table.encode(text, boff)
    t = time.time(); r = bpe._table.encode(text, boff); dt = time.time() - t
    print("BPE host-path %.1f MB in %.4fs = %.1f MB/s (%d tokens)" % (text.size / 1e6, dt, text.size / 1e6 / dt, r[0].size), flush=True)

    # ---- WP encode
    wp = S.FastWP()
    wp.load_resources(os.path.join(G, "ref/resources/pretrained/FastWordPiece"))
    t = time.time(); out = wp.tokenize_batch(pan); dt = time.time() - t
    ok = out == gold["FastWordPiece"]; fails += not ok
    print("WP pan_tadeusz vs golden:", ok, "%.3fs" % dt, wp._trie.stats(), flush=True)
    worc = O.OracleWP(wp._tokens)
    ids, off, st = wp.encode_ids_batch(t5k)
    oids, ooff, ost = worc.tokenize_batch_ids(t5k)
    ok = np.array_equal(ids, oids) and np.array_equal(off, ooff) and np.array_equal(st, ost); fails += not ok
    print("WP train-5K ids vs oracle:", ok, ids.size, oids.size, flush=True)
    fw = ld("fuzz_wp.json")
    texts = [c["text"] for c in fw["sentences"]]
    ids, off, st = wp.encode_ids_batch(texts)
    oids, ooff, ost = worc.tokenize_batch_ids(texts)
    ok = np.array_equal(ids, oids) and np.array_equal(off, ooff) and np.array_equal(st, ost); fails += not ok
    print("WP fuzz vs oracle:", ok, "statuses", np.bincount(st, minlength=3), np.bincount(ost, minlength=3), flush=True)
    if not ok:
        for i in range(len(texts)):
            a = ids[int(off[i]):int(off[i + 1])]; b = oids[int(ooff[i]):int(ooff[i + 1])]
            if st[i] != ost[i] or not np.array_equal(a, b):
                print("   ", i, repr(texts[i][:50]), st[i], ost[i], a[:6], b[:6]); break
    longs = ["słowo " * 3000, "x" * 20000, "", "a", " ".join(pan[:200]), "nie wiem " * 700]
    ids, off, st = wp.encode_ids_batch(longs)
    oids, ooff, ost = worc.tokenize_batch_ids(longs)
    ok = np.array_equal(ids, oids) and np.array_equal(off, ooff) and np.array_equal(st, ost); fails += not ok
    print("WP long/giant:", ok, ids.size, oids.size, st, ost, flush=True)
    wp._trie.encode(text, boff)
    t = time.time(); r = wp._trie.encode(text, boff); dt = time.time() - t
    print("WP host-path %.1f MB in %.4fs = %.1f MB/s (%d tokens)" % (text.size / 1e6, dt, text.size / 1e6 / dt, r[0].size), flush=True)

    # ---- train
    micro = ld("bpe_train_micro.json")
    nb = 0
    for c in micro:
        m = S.NaiveBPE()
        m.train(list(c["corpus"]), c["max_vocab"])
        if [list(p) for p in m.merges_list] != c["merges"] or len(m.vocab) != c["vocab_size"]:
            nb += 1
            if nb <= 3:
                print("   micro mismatch", c["corpus"], c["max_vocab"], m.merges_list, c["merges"])
    fails += bool(nb)
    print("train micro:", len(micro) - nb, "/", len(micro), flush=True)
    g5 = ld("bpe_train5k_1000.json")
    m = S.FastBPE()
    t = time.time(); m.train(t5k, 1000); dt = time.time() - t
    ok = [list(p) for p in m.merges_list] == g5["merges"]; fails += not ok
    print("train-5K 922 merges vs golden:", ok, len(m.merges_list), "%.3fs (reference %.1fs)" % (dt, g5["ref_train_seconds"]), flush=True)
    if not ok:
        for i, (a, b) in enumerate(zip(m.merges_list, g5["merges"])):
            if list(a) != b:
                print("   first diff at", i, a, b); break
    print("FAILS", fails)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
