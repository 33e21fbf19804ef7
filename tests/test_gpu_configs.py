"""BASELINE.json's configs at the sizes they are quoted on (`-m gpu`, one MI355X): configs[2] at 1 M sentences, configs[3] at
2 M word types and as raw text, configs[4]'s per-GPU shard (625 k + 625 k sentences), plus the string-collision replay of the
training loop.  The oracle covers seeded subsamples / the first merges; the full sizes are held by size-independent properties
(batch independence, shard concatenation, recount of the final histogram)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(native):
    if native.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests need an MI355X (there is no CPU fallback to test)")
    native.init(0)
    return native


@pytest.fixture(scope="module")
def synth():
    from subword_tokenizers_amd import synth as S

    return S


@pytest.fixture(scope="module")
def wp30k(swt, dev, synth):
    tok = swt.FastWP()
    tok.vocab = set(synth.v30k())
    tok._build_trie()
    return tok


@pytest.fixture(scope="module")
def bpe8k(swt, dev, synth):
    tok = swt.FastBPE()
    tok.merges_list = list(synth.pretrained_merges()[:8000])
    tok._build_table()
    return tok


def _recount(syms, woff, freq):
    """bpe.py:90-95 over an exported stream, in numpy: {pair key: weighted count}"""
    syms = syms.astype(np.uint64)
    lens = np.diff(woff.astype(np.int64))
    last = np.zeros(syms.size, dtype=bool)
    last[(woff[1:].astype(np.int64) - 1)[lens > 0]] = True
    keys = (syms[:-1] << np.uint64(32)) | syms[1:]
    keep = ~last[:-1]
    w = np.repeat(freq.astype(np.int64), lens)[:-1][keep]
    uk, inv = np.unique(keys[keep], return_inverse=True)
    return uk, np.bincount(inv, weights=w.astype(np.float64)).astype(np.int64)


def _same_histogram(tr):
    syms, woff, freq = tr.export()
    uk, uc = _recount(syms, woff, freq)
    keys, cnts = tr.histogram()
    order = np.argsort(keys)
    assert np.array_equal(keys[order], uk) and np.array_equal(cnts[order].astype(np.int64), uc)


# ------------------------------------------------------------------------------------------------ configs[1], training half

def test_config1_training_to_8000_all_merges(swt, dev, oracle, synth):
    """configs[1]'s training half at spec on the S85k-lex stand-in: FastBPE.train to vocab 8,000 -- EVERY merge (7,9xx of them, most
    of them taken several per step from plateaus of tied pairs, with re-plans and stream squeezes in between), the vocabulary
    and the final symbol stream against the oracle's full recount per merge (bpe.py:88-111)."""
    sents = synth.s85k()
    tok = swt.FastBPE()
    tok.train(sents, 8000)
    orc = oracle.OracleBPETrainer(sents)
    orc.run(8000)
    want = [tuple(m) for m in orc.merges_list]
    got = [tuple(m) for m in tok.merges_list]
    bad = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)
    assert bad is None and len(got) == len(want), (bad, len(got), len(want), got[bad] if bad is not None else None, want[bad] if bad is not None else None)
    assert len(tok.vocab) == orc.vocab_size == 8000
    gs, go, gf = tok._trainer.export()
    ws, wo, wf = orc.export()
    assert np.array_equal(go, wo) and np.array_equal(gf, wf)
    # symbol ids are labels by first appearance of the merged string on both sides
    assert np.array_equal(gs, ws)
    st = tok._trainer.stats()
    assert st["flags"] & 1 == 0  # the index was never abandoned
    _same_histogram(tok._trainer)
    tok.reset()


# ------------------------------------------------------------------------------------------------ configs[2]

def test_config2_wp_one_million_sentences(wp30k, oracle, synth):
    """configs[2] at spec: V30k, 1,000,000 sentences (141 MB).  Oracle on two 10,000-sentence windows (start, 3/4 of the
    batch), statuses, shard-concatenation == whole batch, determinism."""
    n = 1_000_000
    text, off = synth.wp_corpus(n, seed=1000000, vocab=synth.v30k())
    ids, ooff, st = wp30k._trie.encode(text, off)
    assert not st.any() and int(ids.max()) <= len(wp30k._tokens)
    orc = oracle.OracleWP(wp30k._tokens)
    for lo in (0, 750_000):
        sub = synth.unpack(text, off, lo, lo + 10_000)
        oids, oooff, ost = orc.tokenize_batch_ids(sub)
        a, b = int(ooff[lo]), int(ooff[lo + 10_000])
        assert not ost.any()
        assert np.array_equal(ids[a:b], oids) and np.array_equal(ooff[lo:lo + 10_001] - ooff[lo], oooff)
    # two shards of 500 k (what two ranks would encode) concatenate to the whole batch
    cut = int(off[n // 2])
    ia, oa, _ = wp30k._trie.encode(text[:cut].copy(), off[:n // 2 + 1].copy())
    ib, ob, _ = wp30k._trie.encode(text[cut:].copy(), (off[n // 2:] - off[n // 2]).copy())
    assert np.array_equal(np.concatenate([ia, ib]), ids)
    assert np.array_equal(np.concatenate([oa, ob[1:] + oa[-1]]), ooff)
    ids2, ooff2, _ = wp30k._trie.encode(text, off)
    assert np.array_equal(ids2, ids) and np.array_equal(ooff2, ooff)


# ------------------------------------------------------------------------------------------------ configs[3]

def test_config3_two_million_types_first_merges(dev, oracle, synth):
    """configs[3] at spec, reference formulation (bpe.py:73-81: deduplicated word types with frequencies): 2,000,000 types /
    110 M tokens (~1 GiB of text).  First 200 merges (pairs AND counts) and the rewritten stream against the oracle's full
    recount; then 2,000 more merges on the device and the incremental histogram against a recount of the final stream."""
    sym, off, freq = synth.train_words(2_000_000, 1073741824, total_tokens=2_000_000 * 55)
    tr = dev.BpeTrainer.from_words(sym, off, freq)
    lefts, rights, counts = tr.run(200, dev.SYM_BASE)
    orc = oracle.OracleBPETrainer.from_words(sym, off, freq)
    orc.run(10 ** 9, 200)
    ids, cnt = orc.merge_ids()
    assert len(lefts) == 200 == len(ids)
    assert np.array_equal(np.asarray(lefts, dtype=np.uint32), ids[:, 0]) and np.array_equal(np.asarray(rights, dtype=np.uint32), ids[:, 1])
    assert np.array_equal(np.asarray(counts, dtype=np.uint64), cnt)
    got_sym, got_off, got_freq = tr.export()
    want_sym, want_off, want_freq = orc.export()
    assert np.array_equal(got_off, want_off) and np.array_equal(got_sym, want_sym) and np.array_equal(got_freq, want_freq)
    l2, r2, c2 = tr.run(2000, dev.SYM_BASE + 200)
    assert len(l2) == 2000
    assert np.all(np.diff(np.concatenate([counts, c2]).astype(np.int64)) <= 0)  # BPE counts never grow (bpe.py:90-102)
    assert len({(int(a), int(b)) for a, b in zip(np.concatenate([lefts, l2]), np.concatenate([rights, r2]))}) == 2200
    _same_histogram(tr)
    tr.close()


def _raw_text(sym, off, n_tokens, seed, zipf_a=1.05, per_sent=40):
    """space-separated Zipf draws over the word types, a sentence boundary every per_sent words: (uint8 text, uint64 offsets,
    type index of every token)"""
    rng = np.random.default_rng(seed)
    n_types = off.size - 1
    idx = (rng.zipf(zipf_a, size=n_tokens) - 1) % n_types
    cp = sym.astype(np.int64)
    # UTF-8 bytes of every type (the alphabet is Latin/Polish: 1 or 2 bytes per code point)
    two = cp >= 0x80
    assert int(cp.max()) < 0x800
    blen = np.where(two, 2, 1)
    boff = np.zeros(cp.size + 1, dtype=np.int64)
    np.cumsum(blen, out=boff[1:])
    tb = np.zeros(int(boff[-1]), dtype=np.uint8)
    tb[boff[:-1][~two]] = cp[~two]
    tb[boff[:-1][two]] = 0xC0 | (cp[two] >> 6)
    tb[boff[:-1][two] + 1] = 0x80 | (cp[two] & 0x3F)
    wb0 = boff[off[:-1].astype(np.int64)]
    wlen = boff[off[1:].astype(np.int64)] - wb0
    lens = wlen[idx]
    step = lens + 1
    dst = np.cumsum(step) - step
    total = int(step.sum())
    out = np.full(total, 0x20, dtype=np.uint8)
    intra = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
    out[np.repeat(dst, lens) + intra] = tb[np.repeat(wb0[idx], lens) + intra]
    starts = dst[::per_sent]
    soff = np.concatenate([starts, [total]]).astype(np.uint64)
    return out, soff, idx


def test_config3_raw_text_stream(dev, oracle, synth):
    """configs[3] "raw stream" variant: >= 256 MB of text through swt_bpe_train_create_text (device split + Counter, bpe.py:70-81).
    The unique-word stream must be the types in first-occurrence order with their token counts (known from the generator), and
    100 merges from it must equal the oracle's on that word list."""
    sym, off, _ = synth.train_words(400_000, 1073741824, total_tokens=400_000 * 55)
    text, soff, idx = _raw_text(sym, off, 28_000_000, seed=7)
    assert text.size >= 256 * 1000 * 1000
    tr = dev.BpeTrainer.from_text(text, soff)
    got_sym, got_off, got_freq = tr.export()
    # two types may spell the same word: identity is the STRING (bpe.py:76 Counter(words))
    lens = np.diff(off.astype(np.int64))
    words = {}
    canon = np.zeros(off.size - 1, dtype=np.int64)
    for t in np.unique(idx):
        w = sym[int(off[t]):int(off[t + 1])].tobytes()
        canon[t] = words.setdefault(w, t)
    cidx = canon[idx]
    uniq, first, counts = np.unique(cidx, return_index=True, return_counts=True)
    order = np.argsort(first, kind="stable")
    want_types = uniq[order]
    assert got_freq.size == want_types.size
    assert np.array_equal(got_freq.astype(np.int64), counts[order])
    assert np.array_equal(np.diff(got_off.astype(np.int64)), lens[want_types])
    want_sym = np.concatenate([sym[int(off[t]):int(off[t + 1])] for t in want_types[:5000]])
    assert np.array_equal(got_sym[:want_sym.size], want_sym)
    lefts, rights, cnts = tr.run(100, dev.SYM_BASE)
    orc = oracle.OracleBPETrainer.from_words(got_sym, got_off, got_freq)
    orc.run(10 ** 9, 100)
    ids, cnt = orc.merge_ids()
    assert np.array_equal(np.asarray(lefts, dtype=np.uint32), ids[:, 0]) and np.array_equal(np.asarray(rights, dtype=np.uint32), ids[:, 1])
    assert np.array_equal(np.asarray(cnts, dtype=np.uint64), cnt)
    _same_histogram(tr)
    tr.close()


# ------------------------------------------------------------------------------------------------ configs[4]

def test_config4_mixed_shard(bpe8k, wp30k, oracle, synth, dev):
    """configs[4] per GPU: 625,000 sentences through FastBPE (8,000 merges) + 625,000 through FastWP (V30k), one call each,
    as bench.py --workload mixed_encode issues them.  Oracle on a 20,000-sentence window of each half + the status array."""
    n = 625_000
    b_sents = synth.sentences(n, 10000000)
    ids, off = bpe8k.encode_ids_batch(b_sents)
    orc = oracle.OracleBPE(bpe8k.merges_list)
    lo = 300_000
    oids, ooff = orc.tokenize_batch_ids(b_sents[lo:lo + 20_000])
    assert np.array_equal(ids[int(off[lo]):int(off[lo + 20_000])], oids)
    assert np.array_equal(off[lo:lo + 20_001] - off[lo], ooff)
    # the same batch without the word-level dedup gives the same ids
    text, toff = dev.pack_utf8([s.lower() for s in b_sents[:200_000]])
    a, ao = bpe8k._table.encode(text, toff)
    b, bo = bpe8k._table.encode(text, toff, flags=dev.BPE_NO_DEDUP)
    assert np.array_equal(a, b) and np.array_equal(ao, bo) and np.array_equal(a, ids[:int(off[200_000])])
    w_text, w_off = synth.wp_corpus(n, seed=20000000, vocab=synth.v30k())
    wids, woff, wst = wp30k._trie.encode(w_text, w_off)
    assert wst.shape == (n,) and not wst.any()
    worc = oracle.OracleWP(wp30k._tokens)
    sub = synth.unpack(w_text, w_off, lo, lo + 20_000)
    oids, ooff, ost = worc.tokenize_batch_ids(sub)
    assert not ost.any()
    assert np.array_equal(wids[int(woff[lo]):int(woff[lo + 20_000])], oids)
    assert np.array_equal(woff[lo:lo + 20_001] - woff[lo], ooff)


def test_config4_statuses_in_a_large_batch(wp30k, oracle, synth):
    """the status array at batch scale: sentences the reference never returns from, sprinkled into a 200 k-sentence batch"""
    text, off = synth.wp_corpus(200_000, seed=4242, vocab=synth.v30k())
    sents = synth.unpack(text, off)
    bad = {1000: "abc€def", 77_777: "a ## b", 150_001: "İstanbul", 199_999: "x €"}
    for i, s in bad.items():
        sents[i] = s
    ids, ooff, st = wp30k.encode_ids_batch(sents)
    orc = oracle.OracleWP(wp30k._tokens)
    for i in bad:
        hi = min(i + 3, len(sents))
        oi, oo, os_ = orc.tokenize_batch_ids(sents[i - 2:hi])
        assert np.array_equal(st[i - 2:hi], os_) and os_[2] != 0
        assert np.array_equal(ids[int(ooff[i - 2]):int(ooff[hi])], oi)
    assert int(st.astype(bool).sum()) == len(bad)


# ------------------------------------------------------------------------------------------------ collision replay

@pytest.mark.parametrize("cls_name", ["NaiveBPE", "NaiveWP"])
def test_string_collision_replay_branch(swt, dev, oracle, corpora, monkeypatch, cls_name):
    """tokenizers.py: `merged != first + i` (two merges spelling one string, bpe.py:103 / wordpiece.py:96: never observed on real
    data) rebuilds the trainer and re-applies every merge host-driven.  Forced here by slipping a foreign string into the
    symbol table between two interns, which shifts every later id by one: the ids are only labels, so the merges, the
    vocabulary and the final stream must equal the oracle's."""
    from subword_tokenizers_amd import tokenizers as T

    corpus = corpora["pan"][:300]
    calls = {"n": 0, "replays": 0}
    if cls_name == "NaiveBPE":
        real = T._SymbolTable.intern

        def intern(self, s):
            if len(s) > 1 and s not in self.index:
                calls["n"] += 1
                if calls["n"] in (5, 40):
                    real(self, "\x00dummy%d" % calls["n"])
            return real(self, s)

        monkeypatch.setattr(T._SymbolTable, "intern", intern)
        orc = oracle.OracleBPETrainer(corpus)
        target = orc.vocab_size + 90
    else:
        real = T._WpSymbols.intern_merged

        def intern_merged(self, s):
            if s not in self.index:
                calls["n"] += 1
                if calls["n"] in (5, 40):
                    real(self, "\x00dummy%d" % calls["n"])
            return real(self, s)

        monkeypatch.setattr(T._WpSymbols, "intern_merged", intern_merged)
        orc = oracle.OracleWPTrainer(corpus)
        target = orc.vocab_size + 90
    closes = []
    real_close = dev.BpeTrainer.close

    def close(self):
        closes.append(1)
        return real_close(self)

    monkeypatch.setattr(dev.BpeTrainer, "close", close)
    tok = getattr(swt, cls_name)()
    tok.train(list(corpus), target)
    assert len(closes) >= 2, "the replay branch did not run"
    orc.run(target)
    if cls_name == "NaiveBPE":
        assert tok.merges_list == orc.merges_list
    else:
        assert [tuple(p) for p in tok._merge_order] == [tuple(p) for p in orc.merges_list]
    assert len(tok.vocab) == orc.vocab_size == target
    # the replayed device state spells the oracle's stream
    osym, owoff, ofreq = orc.export()
    want = [([orc.symbol(int(x)) for x in osym[int(owoff[w]):int(owoff[w + 1])]], int(ofreq[w])) for w in range(len(owoff) - 1)]
    assert tok.corpus_as_symbols == want
    # and its histogram is the recount of that stream
    _same_histogram(tok._trainer)
