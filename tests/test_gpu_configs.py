"""BASELINE.json's configs at the sizes they are quoted on (`-m gpu`, one MI355X): configs[1] on BOTH stand-ins for train-85k
(the headline workload of bench.py included: S85k-open, every sentence and every merge), configs[2] at 1 M sentences,
configs[3] at 2 M word types to 32,000 merges and as 2^30 bytes of raw text, configs[4]'s per-GPU shard (625 k + 625 k
sentences), plus the string-collision replay of the training loop.  The oracle covers seeded subsamples / the first merges;
the full sizes are held by size-independent properties (batch independence, shard concatenation, recount of the final
histogram).

The three long oracle runs (the C restatement recounts every pair per merge: ~100 s for S85k-open -> 8,000, ~20 s for
S85k-lex, ~60 s for the first 1,000 merges of configs[3]) start in worker threads when the module is set up -- ctypes
releases the GIL around the C call -- and the tests that need them join them, so the suite pays the longest one once."""
import concurrent.futures
import os

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


@pytest.fixture(scope="module")
def oracle_jobs(oracle, synth):
    """name -> Future of a finished oracle trainer"""
    pool = concurrent.futures.ThreadPoolExecutor(max_workers=3)

    def train_text(sents, vocab):
        orc = oracle.OracleBPETrainer(sents)
        orc.run(vocab)
        return orc

    def train_words(n_merges):
        sym, off, freq = synth.train_words(2_000_000, 1073741824, total_tokens=2_000_000 * 55)
        orc = oracle.OracleBPETrainer.from_words(sym, off, freq)
        orc.run(10 ** 9, n_merges)
        return orc

    jobs = {"open": pool.submit(train_text, synth.s85k_open(), 8000),
            "config3": pool.submit(train_words, 1000),
            "lex": pool.submit(train_text, synth.s85k(), 8000)}
    yield jobs
    pool.shutdown(wait=True)


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

def _train_against(swt, sents, orc):
    """FastBPE.train(sents, 8000): EVERY merge, the vocabulary, the final symbol stream and the incremental histogram"""
    tok = swt.FastBPE()
    tok.train(sents, 8000)
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
    return tok


def test_config1_training_to_8000_all_merges(swt, dev, oracle_jobs, synth):
    """configs[1]'s training half at spec on the S85k-lex stand-in: FastBPE.train to vocab 8,000 -- EVERY merge (7,9xx of them, most
    of them taken several per step from plateaus of tied pairs, with re-plans and stream squeezes in between), the vocabulary
    and the final symbol stream against the oracle's full recount per merge (bpe.py:88-111)."""
    _train_against(swt, synth.s85k(), oracle_jobs["lex"].result()).reset()


def test_headline_corpus_encode_all_sentences_three_dedup_modes(bpe8k, dev, oracle, synth):
    """bench.py's headline workload (configs[1] on S85k-open): ALL 85,000 sentences against the oracle, through
    the three paths a handle can take -- adaptive (the handle decides from what the last call saw: two calls, so that both of
    its decisions run), word-level dedup forced, direct -- from device-resident text (what bench.py times) and from list[str]"""
    sents = synth.s85k_open()
    assert len(sents) == 85000
    orc = oracle.OracleBPE(bpe8k.merges_list)
    blob, boff = oracle.pack([s.lower() for s in sents])
    want, woff = orc.tokenize_packed_mt(blob, boff, min(os.cpu_count() or 1, 32))
    text, off = dev.pack_utf8([s.lower() for s in sents])
    try:
        for mode in (dev.DEDUP_AUTO, dev.DEDUP_AUTO, dev.DEDUP_ALWAYS, dev.DEDUP_NEVER):
            bpe8k._table.set_option(dev.OPT_DEDUP, mode)
            ids, ooff = bpe8k._table.encode(text, off)
            assert np.array_equal(ooff, woff), mode
            assert np.array_equal(ids, want), mode
        bpe8k._table.set_option(dev.OPT_DEDUP, dev.DEDUP_AUTO)
        ids, ooff = bpe8k.encode_ids_batch(sents)  # strings -> joined bytes -> device lower + split + encode
        assert np.array_equal(ids, want) and np.array_equal(ooff, woff)
    finally:
        bpe8k._table.set_option(dev.OPT_DEDUP, dev.DEDUP_AUTO)


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

def test_config3_two_million_types_to_32k_merges(dev, oracle_jobs, synth):
    """configs[3] at spec, reference formulation (bpe.py:73-81: deduplicated word types with frequencies):
    2,000,000 types / 110 M tokens (~1 GiB of text) to 32,000 merges.  The first 1,000 merges (pairs AND counts) and the
    rewritten stream against the oracle's full recount per merge; then on to 32,000 on the device, held by what does not depend
    on an oracle: counts never grow (bpe.py:90-102), no pair is merged twice, the inverted index was never abandoned, the
    live-symbol counter equals the exported stream, and the incremental histogram equals a recount of that stream."""
    sym, off, freq = synth.train_words(2_000_000, 1073741824, total_tokens=2_000_000 * 55)
    tr = dev.BpeTrainer.from_words(sym, off, freq)
    lefts, rights, counts = tr.run(1000, dev.SYM_BASE)
    orc = oracle_jobs["config3"].result()
    ids, cnt = orc.merge_ids()
    assert len(lefts) == 1000 == len(ids)
    assert np.array_equal(np.asarray(lefts, dtype=np.uint32), ids[:, 0]) and np.array_equal(np.asarray(rights, dtype=np.uint32), ids[:, 1])
    assert np.array_equal(np.asarray(counts, dtype=np.uint64), cnt)
    got_sym, got_off, got_freq = tr.export()
    want_sym, want_off, want_freq = orc.export()
    assert np.array_equal(got_off, want_off) and np.array_equal(got_sym, want_sym) and np.array_equal(got_freq, want_freq)
    l2, r2, c2 = tr.run(31000, dev.SYM_BASE + 1000)
    assert len(l2) == 31000
    assert np.all(np.diff(np.concatenate([counts, c2]).astype(np.int64)) <= 0)
    pairs = (np.concatenate([lefts, l2]).astype(np.uint64) << np.uint64(32)) | np.concatenate([rights, r2]).astype(np.uint64)
    assert np.unique(pairs).size == 32000
    st = tr.stats()
    assert st["flags"] & 1 == 0
    end_sym, end_off, _ = tr.export()
    assert tr.info()["n_symbols"] == end_sym.size == int(end_off[-1])
    # every merge removed at least one symbol, and symbol ids name the merges in order
    assert int(end_sym.max()) < dev.SYM_BASE + 32000 and end_sym.size <= got_sym.size - 31000
    _same_histogram(tr)
    tr.close()


def _raw_text(sym, off, n_tokens, seed, zipf_a=1.05, per_sent=40, chunk=8_000_000):
    """space-separated Zipf draws over the word types, a sentence boundary every per_sent words: (uint8 text, uint64 offsets,
    type index of every token).  Built in chunks of `chunk` tokens so that 2^30 bytes need a few GB of temporaries, not tens."""
    rng = np.random.default_rng(seed)
    n_types = off.size - 1
    idx = ((rng.zipf(zipf_a, size=n_tokens) - 1) % n_types).astype(np.int32)
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
    total = int(wlen[idx].sum()) + n_tokens  # every token is followed by one space
    out = np.full(total, 0x20, dtype=np.uint8)
    starts = []
    base = 0
    assert chunk % per_sent == 0
    wb0 = wb0.astype(np.int32)
    wlen32 = wlen.astype(np.int32)
    for c0 in range(0, n_tokens, chunk):  # positions inside a chunk fit 32 bits: half the memory traffic of the index arrays
        ix = idx[c0:c0 + chunk]
        lens = wlen32[ix]
        step = lens + 1
        dst = np.cumsum(step, dtype=np.int32) - step
        n_chunk = int(dst[-1]) + int(step[-1])
        intra = np.arange(int(lens.sum()), dtype=np.int32) - np.repeat(np.cumsum(lens, dtype=np.int32) - lens, lens)
        out[base:base + n_chunk][np.repeat(dst, lens) + intra] = tb[np.repeat(wb0[ix], lens) + intra]
        starts.append(dst[::per_sent].astype(np.int64) + base)
        base += n_chunk
    assert base == total
    soff = np.concatenate(starts + [np.array([total], dtype=np.int64)]).astype(np.uint64)
    return out, soff, idx


def test_config3_raw_text_stream(dev, oracle, synth):
    """configs[3] "raw stream" variant at spec ("until 2^30 UTF-8 bytes"): one GiB of text through
    swt_bpe_train_create_text (device split + Counter, bpe.py:70-81).  The unique-word stream must be the types in
    first-occurrence order with their token counts (known from the generator), and 100 merges from it must equal the oracle's
    on that word list."""
    sym, off, _ = synth.train_words(400_000, 1073741824, total_tokens=400_000 * 55)
    text, soff, idx = _raw_text(sym, off, 78_000_000, seed=7)
    assert text.size >= 2 ** 30
    tr = dev.BpeTrainer.from_text(text, soff)
    got_sym, got_off, got_freq = tr.export()
    del text
    # two types may spell the same word: identity is the STRING (bpe.py:76 Counter(words))
    lens = np.diff(off.astype(np.int64))
    n_types = off.size - 1
    seen = np.bincount(idx, minlength=n_types) > 0
    words = {}
    canon = np.arange(n_types, dtype=np.int64)
    for t in np.flatnonzero(seen):
        w = sym[int(off[t]):int(off[t + 1])].tobytes()
        canon[t] = words.setdefault(w, t)
    cidx = canon[idx]
    counts = np.bincount(cidx, minlength=n_types)
    # first occurrence of every type: assigning positions in REVERSE order leaves the earliest one in place
    first = np.full(n_types, -1, dtype=np.int64)
    first[cidx[::-1]] = np.arange(cidx.size - 1, -1, -1, dtype=np.int64)
    uniq = np.flatnonzero(counts)
    assert np.array_equal(cidx[first[uniq]], uniq)
    for t in uniq[[0, uniq.size // 2, uniq.size - 1]]:  # (numpy leaves the LAST write of a duplicate index: checked, not assumed)
        assert first[t] == int(np.argmax(cidx == t))
    order = np.argsort(first[uniq], kind="stable")
    want_types = uniq[order]
    assert got_freq.size == want_types.size
    assert np.array_equal(got_freq.astype(np.int64), counts[want_types])
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


# ------------------------------------------------------------------------------------------------ configs[1], the headline corpus
# (last in the module: the oracle's recount of S85k-open -> 8,000 is the longest of the three background runs)

def test_headline_corpus_training_every_merge(swt, dev, oracle_jobs, synth, golden):
    """bench.py's `train` block at spec: FastBPE.train(S85k-open, 8000) -- EVERY one of the 7,922 merges, the vocabulary, the
    final stream and the histogram against the oracle (bpe.py:88-111), and the three 40-merge windows (start / middle / end)
    against what the REFERENCE'S OWN PYTHON produced there (tests/golden/ref_s85k_open_windows.json, written by
    tools/ref_python_baseline.py in the build container by importing the unmodified class)."""
    tok = _train_against(swt, synth.s85k_open(), oracle_jobs["open"].result())
    ref = golden("ref_s85k_open_windows.json")
    assert ref["max_vocab"] == 8000 and len(ref["windows"]) == 3
    for w in ref["windows"]:
        a = w["first_merge"]
        assert [list(m) for m in tok.merges_list[a:a + len(w["merges"])]] == w["merges"], a
    tok.reset()


@pytest.mark.parametrize("world", [2, 3])
def test_headline_corpus_sharded_fast_path(swt, dev, oracle_jobs, synth, world):
    """the sharded runner at the headline size: S85k-open cut into `world` contiguous sentence ranges, every shard a trainer of
    this process (loop-back communicator: the same kernels and the same exchange protocol as over RCCL, device copies for the
    collectives), to vocab 8,000 -- all 7,922 merges, batches of tied merges ordered by first occurrence over (rank, word,
    offset), against the oracle's unsharded run (bpe.py:88-111)."""
    from subword_tokenizers_amd.distributed import train_sharded_loopback

    want = [tuple(m) for m in oracle_jobs["open"].result().merges_list]
    merges, stats = train_sharded_loopback(synth.s85k_open(), 8000, world)
    bad = next((i for i, (a, b) in enumerate(zip(merges, want)) if a != b), None)
    assert bad is None and len(merges) == len(want), (bad, len(merges), len(want))
    assert all(not (st["flags"] & 1) for st in stats)
    # several merges per step: the fast path ran (one merge per step would need 7,922 steps)
    assert stats[0]["steps"] < 6000, stats[0]

