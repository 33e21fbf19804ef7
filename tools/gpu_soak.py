#!/usr/bin/env python3
"""Differential soak (diagnostics, not part of the suite): many random trials of the device paths against the oracle, new seeds
every run unless SWT_SOAK_SEED is set; prints the seed and the first mismatch.  Training: random word lists over small
alphabets (wide plateaus, twins, chains), whole runs and odd slices.  FastBPE / FastWP: random tables / vocabularies over
tiny alphabets x random texts, through the tile kernels, the single-launch forms, the dedup pipeline and the joined entry."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import subword_tokenizers_amd as S
from subword_tokenizers_amd import _native as N
from oracle import oracle as O

O.build(); O.lib(); N.init(0)
seed = int(os.environ.get("SWT_SOAK_SEED", str(int(time.time()) & 0xFFFFFF)))
budget = float(os.environ.get("SWT_SOAK_SECONDS", "150"))
rng = np.random.default_rng(seed)
print("soak seed", seed, flush=True)
t_end = time.time() + budget
stats = {"train": 0, "train_merges": 0, "bpe": 0, "bpe_tokens": 0, "bpe_proper": 0, "bpe_proper_tokens": 0, "wp": 0, "wp_tokens": 0, "shard": 0, "shard_merges": 0, "wptrain": 0, "wptrain_merges": 0}


def fail(what, **kw):
    print("MISMATCH", what, "seed", seed, {k: (v if not hasattr(v, "tolist") else v.tolist()[:20]) for k, v in kw.items()}, flush=True)
    sys.exit(1)


def train_trial():
    alpha = int(rng.integers(2, 30)); n_words = int(rng.integers(50, 40000)); lo = int(rng.integers(1, 5)); hi = lo + int(rng.integers(1, 14))
    fmax = int(rng.choice([1, 1, 2, 5, 50])); n_merges = int(rng.integers(20, 900))
    lens = rng.integers(lo, hi + 1, size=n_words)
    off = np.zeros(n_words + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
    sym = (97 + rng.integers(0, alpha, size=int(off[-1]))).astype(np.uint32)
    freq = rng.integers(1, fmax + 1, size=n_words).astype(np.uint32)
    orc = O.OracleBPETrainer.from_words(sym, off, freq)
    orc.run(10 ** 9, n_merges)
    ids, cnt = orc.merge_ids()
    slices = None if rng.random() < 0.5 else tuple(int(x) for x in rng.integers(1, 200, size=5))
    tr = N.BpeTrainer.from_words(sym, off, freq)
    ls, rs, cs, i = [], [], [], 0
    while len(ls) < len(ids):
        ask = len(ids) - len(ls) if slices is None else min(slices[i % len(slices)], len(ids) - len(ls))
        l, r, c = tr.run(ask, N.SYM_BASE + len(ls))
        if len(l) != ask: fail("train: short run", alpha=alpha, n_words=n_words, at=len(ls), ask=ask, got=len(l))
        ls += l.tolist(); rs += r.tolist(); cs += c.tolist(); i += 1
    got = np.stack([np.asarray(ls, dtype=np.uint32), np.asarray(rs, dtype=np.uint32)], axis=1) if ls else np.zeros((0, 2), np.uint32)
    if len(ids) and ((got != ids[:, :2]).any() or not np.array_equal(np.asarray(cs, dtype=np.uint64), cnt)):
        bad = int(np.nonzero((got != ids[:, :2]).any(axis=1))[0][0]) if (got != ids[:, :2]).any() else -1
        fail("train: merges", alpha=alpha, n_words=n_words, lo=lo, hi=hi, fmax=fmax, n_merges=n_merges, slices=slices, first_bad=bad)
    gs, go, gf = tr.export(); ws, wo, wf = orc.export()
    if not (np.array_equal(go, wo) and np.array_equal(gs, ws)): fail("train: stream", alpha=alpha, n_words=n_words, n_merges=n_merges)
    tr.close()
    stats["train"] += 1; stats["train_merges"] += len(ids)


ALPHAS = ["ab", "abc", "abcd", "aąb", "ab中", "abcdefgh", "xyżź", "ab-c"]


def texts_over(alpha, n, long_words):
    out = []
    for _ in range(n):
        words = ["".join(alpha[int(c)] for c in rng.integers(0, len(alpha), size=int(rng.integers(1, 200 if long_words else 40))))
                 for _ in range(int(rng.integers(0, 14)))]
        out.append((" " if rng.random() < 0.2 else "") + " ".join(words) + ("." if rng.random() < 0.3 else ""))
    return out


def bpe_trial():
    alpha = ALPHAS[int(rng.integers(len(ALPHAS)))]
    symbols, merges = list(alpha.replace("-", "")), []
    for _ in range(int(rng.integers(2, 80))):
        l, r = symbols[int(rng.integers(len(symbols)))], symbols[int(rng.integers(len(symbols)))]
        if len(l + r) > 30: continue
        merges.append((l, r)); symbols.append(l + r)
    if rng.random() < 0.25 and merges: merges.append(merges[int(rng.integers(len(merges)))])
    tok = S.FastBPE(); tok.merges_list = list(merges); tok._build_table()
    orc = O.OracleBPE(merges)
    texts = texts_over(alpha, int(rng.integers(1, 300)), rng.random() < 0.3)
    modes = [N.DEDUP_NEVER, N.DEDUP_ALWAYS] if rng.random() < 0.5 else [None]
    for mode in modes:
        if mode is not None: tok._table.set_option(N.OPT_DEDUP, mode)
        ids, off = tok.encode_ids_batch(texts)
        oids, ooff = orc.tokenize_batch_ids(texts)
        if not (np.array_equal(off, ooff) and np.array_equal(ids, oids)): fail("bpe batch", alpha=alpha, merges=merges[:50], mode=mode, n=len(texts))
    tok._table.set_option(N.OPT_DEDUP, 0)
    for t in texts[:8]:
        ids, off = tok.encode_ids_batch([t]); oids, ooff = orc.tokenize_batch_ids([t])
        if not (np.array_equal(off, ooff) and np.array_equal(ids, oids)): fail("bpe single", alpha=alpha, merges=merges[:50], text=t)
    stats["bpe"] += 1; stats["bpe_tokens"] += int(oids.size) + int(ids.size)
    tok._table.close()


def wp_trial():
    alpha = ALPHAS[int(rng.integers(len(ALPHAS)))].replace("-", "")
    vocab = set(alpha) | {"##" + c for c in alpha}
    for _ in range(int(rng.integers(0, 120))):
        w = "".join(alpha[int(c)] for c in rng.integers(0, len(alpha), size=int(rng.integers(2, 9))))
        vocab.add(w if rng.random() < 0.4 else "##" + w)
    if rng.random() < 0.3: vocab.discard(alpha[0])           # unknown characters -> UNK paths
    if rng.random() < 0.3: vocab.discard("##" + alpha[-1])
    tok = S.FastWP(); tok.vocab = set(vocab); tok._build_trie()
    orc = O.OracleWP(tok._tokens)
    texts = texts_over(alpha, int(rng.integers(1, 300)), rng.random() < 0.3)
    ids, off, st = tok.encode_ids_batch(texts)
    oids, ooff, ost = orc.tokenize_batch_ids(texts)
    if not (np.array_equal(st, ost) and np.array_equal(off, ooff) and np.array_equal(ids, oids)): fail("wp batch", alpha=alpha, vocab=sorted(vocab)[:80], n=len(texts))
    for t in texts[:6]:
        ids, off, st = tok.encode_ids_batch([t]); oids, ooff, ost = orc.tokenize_batch_ids([t])
        if not (np.array_equal(st, ost) and np.array_equal(off, ooff) and np.array_equal(ids, oids)): fail("wp single", alpha=alpha, vocab=sorted(vocab)[:80], text=t)
    stats["wp"] += 1; stats["wp_tokens"] += int(oids.size)
    tok._trie.close()


def shard_trial():
    """the sharded runner (loop-back: every shard a trainer of this process) in its fast and its generic form: a random word
    list cut into 2..5 contiguous ranges (some empty), merges and counts against the oracle on the whole list"""
    alpha = int(rng.integers(2, 12)); n_words = int(rng.integers(20, 6000)); lo = int(rng.integers(1, 4)); hi = lo + int(rng.integers(1, 9))
    fmax = int(rng.choice([1, 1, 2, 9])); n_merges = int(rng.integers(10, 400)); world = int(rng.integers(2, 6))
    lens = rng.integers(lo, hi + 1, size=n_words)
    off = np.zeros(n_words + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
    sym = (97 + rng.integers(0, alpha, size=int(off[-1]))).astype(np.uint32)
    freq = rng.integers(1, fmax + 1, size=n_words).astype(np.uint32)
    orc = O.OracleBPETrainer.from_words(sym, off, freq)
    orc.run(10 ** 9, n_merges)
    ids, cnt = orc.merge_ids()
    cuts = sorted(int(x) for x in rng.integers(0, n_words + 1, size=world - 1))
    bounds = [0] + cuts + [n_words]
    os.environ["SWT_DIST_GENERIC"] = "1" if rng.random() < 0.25 else "0"
    trainers = []
    for r in range(world):
        a, b = bounds[r], bounds[r + 1]
        trainers.append(N.BpeTrainer.from_words(sym[int(off[a]):int(off[b])], off[a:b + 1] - off[a], freq[a:b]))
    comm = N.Dist.loopback(world)
    try:
        comm.shard_begin(trainers)
        ls, rs, cs = [], [], []
        slices = None if rng.random() < 0.5 else tuple(int(x) for x in rng.integers(1, 120, size=4))
        i = 0
        while len(ls) < len(ids):
            ask = len(ids) - len(ls) if slices is None else min(slices[i % len(slices)], len(ids) - len(ls))
            l, r, c = comm.run(trainers, ask, N.SYM_BASE + len(ls))
            if len(l) != ask: fail("shard: short run", world=world, bounds=bounds, at=len(ls), ask=ask, got=len(l))
            ls += l.tolist(); rs += r.tolist(); cs += c.tolist(); i += 1
        got = np.stack([np.asarray(ls, dtype=np.uint32), np.asarray(rs, dtype=np.uint32)], axis=1) if ls else np.zeros((0, 2), np.uint32)
        if len(ids) and ((got != ids[:, :2]).any() or not np.array_equal(np.asarray(cs, dtype=np.uint64), cnt)):
            bad = int(np.nonzero((got != ids[:, :2]).any(axis=1))[0][0]) if (got != ids[:, :2]).any() else -1
            fail("shard: merges", alpha=alpha, n_words=n_words, lo=lo, hi=hi, fmax=fmax, n_merges=n_merges, world=world, bounds=bounds,
                 generic=os.environ["SWT_DIST_GENERIC"], slices=slices, first_bad=bad)
    finally:
        for t in trainers: t.close()
        comm.close()
    stats["shard"] += 1; stats["shard_merges"] += len(ids)


def wptrain_trial():
    """NaiveWP.train (fused step while the list of live pairs is short, the generic four launches with SWT_WP_GENERIC) against
    the oracle's merge order on random sentences over a small alphabet"""
    alpha = ALPHAS[int(rng.integers(len(ALPHAS)))].replace("-", "")
    texts = texts_over(alpha, int(rng.integers(70, 400)), False)
    if rng.random() < 0.3: os.environ["SWT_WP_GENERIC"] = "1"
    else: os.environ.pop("SWT_WP_GENERIC", None)
    orc = O.OracleWPTrainer(texts)
    target = orc.vocab_size + int(rng.integers(5, 250))
    orc.run(target)
    tok = S.NaiveWP(); tok.train(list(texts), target)
    got = [tuple(m) for m in tok._merge_order]; want = [tuple(m) for m in orc.merges_list]
    if got != want:
        bad = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
        fail("wp train: merge order", alpha=alpha, n=len(texts), target=target, first_bad=bad, generic=os.environ.get("SWT_WP_GENERIC"))
    tok.reset()
    os.environ.pop("SWT_WP_GENERIC", None)
    stats["wptrain"] += 1; stats["wptrain_merges"] += len(want)


def bpe_proper_trial():
    """PROPER tables (every pair ranks above the merges that make its symbols: every merged string is made once) take the
    word-lane kernel's fast path -- live-slot mask, lane refill, four lanes a word in the tail; texts with long words, many
    short words per chunk, multi-byte symbols, spans over several chunks; direct and dedup paths, and single sentences."""
    import ctypes as C
    alpha = ALPHAS[int(rng.integers(len(ALPHAS)))]
    base = list(alpha.replace("-", ""))
    symbols, have, merges = list(base), set(base), []
    for _ in range(int(rng.integers(3, 120))):
        # prefer recent symbols so that chains of merges (long rounds) exist
        pool = symbols[-12:] if rng.random() < 0.6 else symbols
        l, r = pool[int(rng.integers(len(pool)))], symbols[int(rng.integers(len(symbols)))]
        if len(l + r) > 28 or (l + r) in have or (l, r) in merges: continue
        merges.append((l, r)); symbols.append(l + r); have.add(l + r)
    if not merges: return
    tok = S.FastBPE(); tok.merges_list = list(merges); tok._build_table()
    lib = N.lib()
    lib.swt_debug_bpe_table_info.restype = C.c_int
    lib.swt_debug_bpe_table_info.argtypes = [C.c_void_p, C.c_int]
    if lib.swt_debug_bpe_table_info(tok._table._h, 2) != 1: fail("bpe proper: the table was not recognised as proper", merges=merges[:40])
    orc = O.OracleBPE(merges)
    n_sent = int(rng.integers(1, 400))
    shape = int(rng.integers(4))
    texts = []
    for _ in range(n_sent):
        if shape == 0:    # many short words
            words = ["".join(base[int(c)] for c in rng.integers(0, len(base), size=int(rng.integers(1, 5)))) for _ in range(int(rng.integers(0, 120)))]
        elif shape == 1:  # words around the 32-slot mask and beyond
            words = ["".join(base[int(c)] for c in rng.integers(0, len(base), size=int(rng.integers(20, 70)))) for _ in range(int(rng.integers(0, 12)))]
        elif shape == 2:  # built from the table's own symbols: long chains of merges
            words = ["".join(symbols[int(c)] for c in rng.integers(0, len(symbols), size=int(rng.integers(1, 6)))) for _ in range(int(rng.integers(0, 40)))]
        else:
            words = ["".join(base[int(c)] for c in rng.integers(0, len(base), size=int(rng.integers(1, 40)))) for _ in range(int(rng.integers(0, 30)))]
        sep = "." if rng.random() < 0.2 else " "
        texts.append((" " if rng.random() < 0.1 else "") + sep.join(words) + ("!" if rng.random() < 0.3 else ""))
    oids, ooff = orc.tokenize_batch_ids(texts)
    for mode in (N.DEDUP_NEVER, N.DEDUP_ALWAYS):
        tok._table.set_option(N.OPT_DEDUP, mode)
        ids, off = tok.encode_ids_batch(texts)
        if not (np.array_equal(off, ooff) and np.array_equal(ids, oids)):
            bad = next((i for i in range(len(texts)) if off[i + 1] != ooff[i + 1] or not np.array_equal(ids[int(off[i]):int(off[i + 1])], oids[int(ooff[i]):int(ooff[i + 1])])), -1)
            fail("bpe proper batch", alpha=alpha, merges=merges[:60], mode=mode, n=len(texts), shape=shape, first_bad=bad, text=texts[bad][:200] if bad >= 0 else None)
    tok._table.set_option(N.OPT_DEDUP, 0)
    for t in texts[:6]:
        ids, off = tok.encode_ids_batch([t]); o1, oo1 = orc.tokenize_batch_ids([t])
        if not (np.array_equal(off, oo1) and np.array_equal(ids, o1)): fail("bpe proper single", alpha=alpha, merges=merges[:60], text=t[:200])
    stats["bpe_proper"] += 1; stats["bpe_proper_tokens"] += int(oids.size)
    tok._table.close()


only = os.environ.get("SWT_SOAK_ONLY")
trials = (train_trial, bpe_trial, wp_trial, shard_trial, wptrain_trial, bpe_proper_trial)
if only: trials = tuple(f for f in trials if f.__name__ in only.split(","))
k = 0
while time.time() < t_end:
    trials[k % len(trials)]()
    k += 1
    if k % 30 == 0: print(k, stats, flush=True)
print("soak ok: seed", seed, stats, flush=True)
