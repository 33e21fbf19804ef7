"""Parity of the HIP path (through the C ABI of libswt_hip.so) against the CPU oracle and the committed golden
vectors.  Needs a real MI355X: run with `-m gpu`.  Bit-exact everywhere: this is integer/index work."""
import contextlib
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
def bpe(swt, dev, ref_dir):
    tok = swt.FastBPE()
    tok.load_resources(os.path.join(ref_dir, "resources/pretrained/FastBPE"))
    return tok


@pytest.fixture(scope="module")
def bpe_orc(oracle, bpe):
    return oracle.OracleBPE(bpe.merges_list)


@pytest.fixture(scope="module")
def wp(swt, dev, ref_dir):
    tok = swt.FastWP()
    tok.load_resources(os.path.join(ref_dir, "resources/pretrained/FastWordPiece"))
    return tok


@pytest.fixture(scope="module")
def wp_orc(oracle, wp):
    return oracle.OracleWP(wp._tokens)


@contextlib.contextmanager
def dedup(dev, mode, *toks, bits=0):
    """SWT_OPT_DEDUP (and the word table's size) on the handles of these tokenizers, restored afterwards"""
    handles = [getattr(t, "_table", None) or t._trie for t in toks]
    try:
        for h in handles:
            h.set_option(dev.OPT_DEDUP, mode)
            h.set_option(dev.OPT_DEDUP_TABLE_BITS, bits)
        yield
    finally:
        for h in handles:
            h.set_option(dev.OPT_DEDUP, dev.DEDUP_AUTO)
            h.set_option(dev.OPT_DEDUP_TABLE_BITS, 0)


def same_bpe(tok, orc, texts):
    ids, off = tok.encode_ids_batch(texts)
    oids, ooff = orc.tokenize_batch_ids(texts)
    assert np.array_equal(off, ooff)
    assert np.array_equal(ids, oids)
    return ids, off


def same_wp(tok, orc, texts):
    ids, off, st = tok.encode_ids_batch(texts)
    oids, ooff, ost = orc.tokenize_batch_ids(texts)
    assert np.array_equal(st, ost)
    assert np.array_equal(off, ooff)
    assert np.array_equal(ids, oids)
    return ids, off, st


# ---------------------------------------------------------------------------------------------- BPE encode

def test_bpe_author_golden(bpe, corpora):
    """reference data/pan_tadeusz.tokens.json (989 sentences, 11,117 tokens)"""
    out = bpe.tokenize_batch(corpora["pan"])
    assert out == corpora["pan_tokens"]["FastBPE"]
    assert sum(map(len, out)) == 11117
    assert bpe.tokenize(corpora["pan"][3]) == corpora["pan_tokens"]["FastBPE"][3]  # the reference's one-sentence call


def test_bpe_train5k_ids(bpe, bpe_orc, corpora):
    ids, off = same_bpe(bpe, bpe_orc, corpora["t5k"])
    assert ids.size == 101863


def test_bpe_fuzz_golden(swt, bpe, golden):
    fz = golden("fuzz_bpe.json")
    texts = [c["text"] for c in fz["sentences"]]
    out = bpe.tokenize_batch(texts)
    for c, got in zip(fz["sentences"], out):
        assert got == c["pretrained"], repr(c["text"])
    t5 = swt.FastBPE()
    t5.merges_list = [tuple(m) for m in golden("bpe_train5k_1000.json")["merges"]]
    t5._build_table()
    out = t5.tokenize_batch(texts)
    for c, got in zip(fz["sentences"], out):
        assert got == c["t5k"], repr(c["text"])
    for c in fz["encode_word"]:
        assert bpe.encode_word(c["word"]) == c["pretrained"]
        assert t5.encode_word(c["word"]) == c["t5k"]


def test_bpe_encode_word_is_not_pretokenized(bpe, bpe_orc):
    """encode_word takes the string as ONE word (bpe.py:206): spaces and punctuation are ordinary symbols"""
    for w in ["a b", "nie wiem, co", "x.y", " ", "ab" * 40, "słowo słowo"]:
        assert bpe.encode_word(w) == bpe_orc.encode_word(w), repr(w)
    assert bpe.encode_word("") == [""]


def test_bpe_edge_shapes(bpe, bpe_orc):
    cases = [
        [],
        [""],
        ["", "", ""],
        ["a"],
        ["", "a", "", "b", ""],
        [" ", "  ", "\t\n"],
        ["!" * 5000],                      # every byte its own word
        ["słowo " * 3000],                 # one sentence spanning many chunks
        ["x" * 20000],                     # a single word longer than the LDS chunk (global-memory path)
        ["ab" * 6000 + " " + "nie " * 10],
        ["ż" * 3000 + " koniec"],          # multi-byte giant word
        ["a" * 4095, "b" * 4096, "c" * 4097, "d" * 2047, "e" * 2048, "f" * 2049],  # around tile/chunk sizes
        ["wyraz"] * 3000,                  # many short sentences per tile
        [("zdanie numer %d. " % i) * (i % 7) for i in range(500)],
        ["\U0001F600 emoji \U0001F600\U0001F601 x", "中文 字", "İstanbul", "áb", "a\x00b", "\ud800x"],
    ]
    for texts in cases:
        same_bpe(bpe, bpe_orc, texts)


def test_bpe_dedup_path_equals_direct_path(bpe, bpe_orc, dev, corpora):
    """word-level dedup inside a call (default for batches >= 1.75 MiB) and the direct path give the same ids"""
    cases = [
        [], [""], ["", "a", "", "b", ""], ["!" * 5000], ["słowo " * 3000], ["x" * 20000], ["ab" * 6000 + " " + "nie " * 10],
        ["ż" * 3000 + " koniec"], ["q" * 300 + " " + "q" * 300, "q" * 300],          # 255+-byte words are never matched
        ["a" * 4095, "b" * 4096, "c" * 4097, "d" * 2047, "e" * 2048, "f" * 2049, "g" * 1023, "h" * 1024, "i" * 1025],
        ["wyraz"] * 3000, corpora["pan"][:400], ["\U0001F600 emoji \U0001F600\U0001F601 x", "中文 字", "İstanbul", "a\x00b"],
    ]
    with dedup(dev, dev.DEDUP_ALWAYS, bpe):  # dedup whatever the size
        for texts in cases:
            same_bpe(bpe, bpe_orc, texts)
        ids_d, off_d = bpe.encode_ids_batch(corpora["t5k"])
    with dedup(dev, dev.DEDUP_NEVER, bpe):
        ids_n, off_n = bpe.encode_ids_batch(corpora["t5k"])
    assert np.array_equal(ids_d, ids_n) and np.array_equal(off_d, off_n)
    # calling again reuses the table under a new epoch
    for _ in range(3):
        ids_e, off_e = bpe.encode_ids_batch(corpora["t5k"])
        assert np.array_equal(ids_e, ids_n) and np.array_equal(off_e, off_n)


def test_bpe_dedup_growing_batches_and_epoch_wrap(swt, oracle, dev, bpe, corpora):
    """a fresh handle whose workspaces grow from call to call (scan tickets of new buffers), and more than 256 calls on
    one handle (the word table's 8-bit epoch wraps and the table is cleared)"""
    merges = list(bpe.merges_list[:2000])
    tok = swt.FastBPE()
    tok.merges_list = list(merges)
    tok._build_table()
    orc = oracle.OracleBPE(merges)
    pan, t5k = corpora["pan"], corpora["t5k"]
    with dedup(dev, dev.DEDUP_ALWAYS, tok):
        for texts in (pan[:3], pan[:200], t5k[:3000], pan[:50], t5k, ["a b c"]):
            same_bpe(tok, orc, texts)
        small = pan[:40]
        want = orc.tokenize_batch_ids(small)
        for i in range(300):
            ids, off = tok.encode_ids_batch(small)
            assert np.array_equal(ids, want[0]) and np.array_equal(off, want[1]), i


def test_dedup_table_overflow_is_harmless(swt, dev, bpe, bpe_orc, wp, wp_orc, corpora):
    """the word table is small on purpose; a word that finds no slot is encoded on its own -- same ids whatever the load"""
    texts = corpora["t5k"][:2500] + ["x" * 300 + " " + "x" * 300, "hello! ok", "a ## b"]
    for bits in (4, 8, 12):
        with dedup(dev, dev.DEDUP_ALWAYS, bpe, wp, bits=bits):
            same_bpe(bpe, bpe_orc, texts)
            same_wp(wp, wp_orc, texts)
    same_bpe(bpe, bpe_orc, texts)
    same_wp(wp, wp_orc, texts)


def test_single_launch_direct_mode_boundaries(bpe, bpe_orc, wp, wp_orc):
    """tiny calls run as one launch of one workgroup that writes the caller's arrays itself (DirectOut): sizes and sentence
    counts on both sides of its limits (1 KiB / 2 KiB of text, 64 sentences), chunk cuts and giant words inside it"""
    w = "słowo "
    cases = [
        ["a"], ["", "a", ""], ["x" * 1024], ["x" * 1025], ["y" * 600 + " " + "z" * 423], ["y" * 600 + " " + "z" * 424],
        [w * 170], [w * 171], ["ab " * 341], ["ab " * 342], ["q" * 2048], ["q" * 2049], [("nie wiem, " * 204)[:2048]], [("nie wiem, " * 205)[:2049]],
        ["a"] * 64, ["a"] * 65, ["ab cd"] * 64, [""] * 64, [""] * 65, ["hello!", "ok"], ["a ## b"] * 3, ["zażółć gęślą jaźń!"] * 40,
    ]
    for texts in cases:
        same_bpe(bpe, bpe_orc, texts)
        same_wp(wp, wp_orc, texts)
    for texts in cases:  # and one sentence per call, the reference's own way
        for t in texts[:3]:
            assert bpe.tokenize(t) == bpe_orc.tokenize(t)


def test_random_tables_and_texts_both_paths(swt, oracle, dev):
    """seeded stress: random merge tables / vocabularies over tiny alphabets (twin runs, overlapping merges, punctuation and
    '#' inside chunks), random ragged texts; direct path and dedup path against the oracle"""
    import random
    rng = random.Random(20261004)
    for case in range(12):
        alpha = rng.choice(["ab", "abc", "aąb", "abcdż", "xy#", "ab.,"])
        letters = [c for c in alpha if c.isalnum()] or ["a"]
        # a BPE table: random pairs over symbols that exist when the merge is made
        syms = list(dict.fromkeys(letters))
        merges = []
        for _ in range(rng.randint(1, 25)):
            l, r = rng.choice(syms), rng.choice(syms)
            if (l, r) not in merges:
                merges.append((l, r))
                syms.append(l + r)
        bpe = swt.FastBPE()
        bpe.merges_list = list(merges)
        bpe._build_table()
        borc = oracle.OracleBPE(merges)
        # a WordPiece vocabulary: single characters (so that the reference terminates) + random pieces
        pieces = set(alpha) | {"##" + c for c in alpha if c != "#"}
        for _ in range(rng.randint(0, 20)):
            w = "".join(rng.choice(letters) for _ in range(rng.randint(2, 5)))
            pieces.add(w if rng.random() < 0.5 else "##" + w)
        wp = swt.FastWP()
        wp.vocab = set(pieces)
        wp._build_trie()
        worc = oracle.OracleWP(wp._tokens)
        texts = []
        for _ in range(rng.randint(1, 60)):
            words = ["".join(rng.choice(alpha) for _ in range(rng.randint(1, 12))) for _ in range(rng.randint(0, 30))]
            texts.append(rng.choice(["", " ", "  "]).join([""] + words) if rng.random() < 0.2 else " ".join(words))
        for mode in (dev.DEDUP_NEVER, dev.DEDUP_ALWAYS):
            with dedup(dev, mode, bpe, wp):
                same_bpe(bpe, borc, texts)
                same_wp(wp, worc, texts)


def test_device_lowercase_equals_str_lower(swt, dev, bpe, bpe_orc, corpora):
    """swt_utf8_lower + the host splice of flagged sentences == [t.lower() for t in texts], byte for byte"""
    import random
    rng = random.Random(248)
    alphabet = ("ABCXYZabc ĄĆĘŁŃÓŚŹŻ ÀÉÎÕÜ ΑΒΓΣΩσς ЖЩЯ İI ẞ K Ω Å ȺȾ ⱢⱤ Ꞔ \U00010400\U0001E900 ǅ Ǆ ǲ 中文 \x00\x1c ,.;!?\u00a0\u2028")
    texts = ["", "A", "ŻÓŁĆ gęślą JAŹŃ", "ΟΔΥΣΣΕΥΣ ΣΑΣ Σ", "İstanbul İİ", "STRAẞE", "K", "x" * 5000 + "Ω", "Z" * 70000]
    for _ in range(300):
        texts.append("".join(rng.choice(alphabet) for _ in range(rng.randint(0, 80))))
    texts += [t.upper() for t in corpora["pan"][:300]] + corpora["t5k"][:500]
    got_text, got_off = dev.pack_and_lower(texts)
    want_text, want_off = dev.pack_utf8([t.lower() for t in texts])
    assert np.array_equal(got_off, want_off) and np.array_equal(got_text, want_text)
    # and through the encoder: upper-case input gives the ids of the lowercased text
    same_bpe(bpe, bpe_orc, [t.upper() for t in corpora["pan"][:200]] + ["ΣΟΦΟΣ İstanbul STRAẞE"])


def test_bpe_ragged_random_batches(bpe, bpe_orc, corpora):
    rng = np.random.default_rng(7)
    pool = corpora["t5k"] + corpora["pan"] + ["", " ", "x" * 5000, "ala, ma! kota?"]
    for _ in range(6):
        k = int(rng.integers(1, 400))
        texts = [pool[int(i)] for i in rng.integers(0, len(pool), size=k)]
        same_bpe(bpe, bpe_orc, texts)


def test_bpe_duplicate_and_unreachable_merges(swt, oracle, dev):
    merges = [("a", "b"), ("c", "d"), ("a", "b"), ("ab", "cd"), ("xy", "z"), ("b", "c")]
    tok = swt.FastBPE()
    tok.merges_list = list(merges)
    tok._build_table()
    orc = oracle.OracleBPE(merges)
    for w in ["abcd", "abcdabcd", "xyz", "bcbc", "aabbccdd"]:
        assert tok.tokenize(w) == orc.tokenize(w), w


def test_joined_entry_points_and_their_fallbacks(swt, dev, bpe, bpe_orc, wp, wp_orc, corpora):
    """encode_ids_batch on more than 64 texts: strings -> join_texts -> swt_bpe_encode_joined / swt_wp_encode_joined (the
    prepared text stays on the device); texts that hold U+0000, texts only str.lower() lowercases and all-empty batches take
    the long way (pack_and_lower -> swt_*_encode) -- same ids either way"""
    base = corpora["pan"][:150]
    batches = [
        base,
        [t.upper() for t in base],
        base[:70] + ["a\x00b c"] + base[70:],                    # U+0000 inside a text
        base[:30] + ["İstanbul ΣΟΦΟΣ STRAẞE"] + base[30:],         # need_host: the device flags the sentence
        [""] * 100,
        ["", "x"] * 60,
        ["\ud800 lone", "\U0001F600 emoji"] + base[:80] + ["中文 字", "é"],
        ["ł" + "a" * 7, "a" * 7 + "ł", "a" * 8 + "ł", "ł" * 8] * 20,
    ]
    for texts in batches:
        same_bpe(bpe, bpe_orc, texts)
        same_wp(wp, wp_orc, texts)
    # the entry points themselves: ids when the device can lowercase everything, None when a sentence needs the host
    joined, n_nul = dev.join_texts(base)
    assert n_nul == 0
    ids, off = bpe._table.encode_joined(joined, len(base))
    oids, ooff = bpe_orc.tokenize_batch_ids(base)
    assert np.array_equal(ids, oids) and np.array_equal(off, ooff)
    joined, _ = dev.join_texts(base[:30] + ["İ"] + base[30:])
    assert bpe._table.encode_joined(joined, len(base) + 1) is None
    assert wp._trie.encode_joined(joined, len(base) + 1) is None
    with pytest.raises(TypeError):
        bpe.encode_ids_batch(base + [3])
    with pytest.raises(TypeError):
        wp.encode_ids_batch(base + [b"x"])
    # a wrong sentence count is refused, not mis-split
    with pytest.raises(dev.SwtError):
        bpe._table.encode_joined(dev.join_texts(base)[0], len(base) + 5)


def test_bpe_twin_runs_across_register_sets(swt, oracle, dev):
    """runs of one symbol ("aaaa...": bpe.py:225-235 merges them left to right, non-overlapping) that start anywhere in the
    list of a chunk, longer than a wave and across the 64-entry sets of the register rounds"""
    merges = [("a", "a"), ("aa", "aa"), ("b", "b"), ("a", "b"), ("aaaa", "aa"), ("bb", "b"), ("c", "a"), ("aa", "a")]
    tok = swt.FastBPE()
    tok.merges_list = list(merges)
    tok._build_table()
    orc = oracle.OracleBPE(merges)
    texts = []
    for n in range(1, 190, 3):
        texts.append("ba " * (n % 7) + "a" * n + " " + "a" * max(1, 150 - n) + "b" * (n % 4))
        texts.append("ca" * (n % 11) + " " + "b" * n + "a" * (n // 2) + " c" + "a" * (n % 67))
    texts += ["a" * 63 + " " + "a" * 64 + " " + "a" * 65, "ab " * 20 + "a" * 127, "b" * 3 + " " + "a" * 128 + " " + "a" * 129,
              "a" * 250, "aaa " * 60, "a" * 191 + " " + "a" * 192 + " " + "a" * 193]
    same_bpe(tok, orc, texts)
    for t in texts[:40]:
        same_bpe(tok, orc, [t])


def test_bpe_random_tables_fuzz(swt, oracle, dev):
    """random merge tables over tiny alphabets (so that words are long chains of mergeable pairs, twin runs and overlapping
    candidates) x random texts, through the tile kernel, the single-launch form and the dedup pipeline"""
    rng = np.random.default_rng(2024)
    for trial in range(24):
        alpha = ["ab", "abc", "abcd", "aąb", "ab\u4e2d"][trial % 5]
        symbols = list(alpha)
        merges = []
        for _ in range(int(rng.integers(3, 40))):
            l, r = symbols[int(rng.integers(len(symbols)))], symbols[int(rng.integers(len(symbols)))]
            if len(l + r) > 24:
                continue
            merges.append((l, r))
            symbols.append(l + r)
        if trial % 4 == 0 and merges:
            merges.append(merges[0])  # a duplicate: the last rank wins (bpe.py:200)
        tok = swt.FastBPE()
        tok.merges_list = list(merges)
        tok._build_table()
        orc = oracle.OracleBPE(merges)
        texts = []
        for _ in range(120):
            n_words = int(rng.integers(0, 12))
            words = ["".join(alpha[int(c)] for c in rng.integers(0, len(alpha), size=int(rng.integers(1, 60 if trial % 3 else 200))))
                     for _ in range(n_words)]
            texts.append((" " if rng.random() < 0.2 else "") + " ".join(words) + ("." if rng.random() < 0.3 else ""))
        same_bpe(tok, orc, texts)
        for t in texts[:12]:
            same_bpe(tok, orc, [t])
            assert tok.encode_word(t.replace(" ", "")[:150] or "a") == orc.encode_word(t.replace(" ", "")[:150] or "a")
        with dedup(dev, dev.DEDUP_ALWAYS, tok):
            same_bpe(tok, orc, texts)
        tok._table.close()


def test_bpe_wide_table_unpacked_path(swt, oracle, dev, bpe, corpora):
    """more than 65,534 merges: the kernel variant whose cached pair value is the bare rank (merged_of_rank[] path)"""
    extra = [(chr(0xE000 + 2 * i), chr(0xE001 + 2 * i)) for i in range(3000)]          # private-use pairs, never in text
    extra += [(chr(0x4E00 + (i % 20000)), chr(0x3400 + (i // 20000))) for i in range(48000)]
    merges = list(bpe.merges_list) + extra
    assert len(merges) > 65534
    tok = swt.FastBPE()
    tok.merges_list = merges
    tok._build_table()
    orc = oracle.OracleBPE(merges)
    texts = corpora["pan"][:300] + ["\ue000\ue001 x", "\u4e00\u3400\u4e01\u3400", "zażółć " * 50]
    same_bpe(tok, orc, texts)
    # same ids as the packed table on ordinary text (the extra merges never fire there)
    a, ao = tok.encode_ids_batch(corpora["pan"][:300])
    b, bo = bpe.encode_ids_batch(corpora["pan"][:300])
    assert np.array_equal(ao, bo) and [tok.decode_ids(a)] == [bpe.decode_ids(b)]


def test_bpe_s85k_full_size_properties(swt, oracle, dev):
    """config 2 at full size: bit-exact vs the oracle on ALL of S85k, plus size-independent properties"""
    from subword_tokenizers_amd import synth

    sents = synth.s85k()
    tok = swt.FastBPE()
    tok.merges_list = list(synth.pretrained_merges()[:8000])
    tok._build_table()
    orc = oracle.OracleBPE(tok.merges_list)
    ids, off = same_bpe(tok, orc, sents)
    # determinism and batch-independence: a different batching gives the same per-sentence ids
    ids2, off2 = tok.encode_ids_batch(sents[::-1])
    n = len(sents)
    for i in (0, 1, n // 2, n - 1):
        a = ids[int(off[i]):int(off[i + 1])]
        b = ids2[int(off2[n - 1 - i]):int(off2[n - i])]
        assert np.array_equal(a, b)
    # detokenisation round trip: concatenated token strings spell the pre-tokenized words
    words = [w for w, _ in tok.preprocessing([sents[5]])[0]]
    toks = tok.decode_ids(ids[int(off[5]):int(off[6])])
    assert "".join(t[2:] if t.startswith("##") else t for t in toks) == "".join(words)
    assert sum(not t.startswith("##") for t in toks) == len(words)


def test_bpe_device_buffer_entry_point(bpe, bpe_orc, dev, corpora):
    """swt_bpe_encode_dev with caller-owned device buffers (torch is only the allocator here)"""
    torch = pytest.importorskip("torch")
    texts = corpora["pan"][:300]
    text, off = dev.pack_utf8([t.lower() for t in texts])
    d_text = torch.from_numpy(text.copy()).cuda()
    d_off = torch.from_numpy(off.view(np.int64).copy()).cuda()
    d_out = torch.empty(text.size + 64, dtype=torch.int32, device="cuda")
    d_out_off = torch.empty(len(texts) + 1, dtype=torch.int64, device="cuda")
    d_n = torch.zeros(1, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        bpe._table.encode_dev(d_text.data_ptr(), int(text.size), d_off.data_ptr(), len(texts), d_out.data_ptr(),
                              d_out_off.data_ptr(), d_n.data_ptr(), 0, side.cuda_stream)
    side.synchronize()
    n = int(d_n.item())
    oids, ooff = bpe_orc.tokenize_batch_ids(texts)
    assert np.array_equal(d_out[:n].cpu().numpy().view(np.uint32), oids)
    assert np.array_equal(d_out_off.cpu().numpy().view(np.uint64), ooff)


# ---------------------------------------------------------------------------------------------- WP encode

def test_wp_author_golden(wp, corpora):
    out = wp.tokenize_batch(corpora["pan"])
    assert out == corpora["pan_tokens"]["FastWordPiece"]
    assert sum(map(len, out)) == 29164
    assert wp.tokenize(corpora["pan"][3]) == corpora["pan_tokens"]["FastWordPiece"][3]


def test_wp_train5k_ids(wp, wp_orc, corpora):
    ids, off, st = same_wp(wp, wp_orc, corpora["t5k"])
    assert ids.size == 312665 and not st.any()  # zero UNK-free, all terminate


def test_wp_fuzz_golden_with_nontermination(swt, wp, golden, ref_dir):
    fw = golden("fuzz_wp.json")
    tut = swt.FastWP()
    tut.load_resources(os.path.join(ref_dir, "resources/tests/FastWordPiece"))

    def run(tok, text):
        try:
            return tok.tokenize(text)
        except RuntimeError:
            return "TIMEOUT"  # the reference spins forever; we refuse (documented deviation)
        except IndexError:
            return "INDEXERROR"

    for c in fw["sentences"]:
        assert run(wp, c["text"]) == c["pretrained"], repr(c["text"])
        if c["tutorial"] is not None:
            assert run(tut, c["text"]) == c["tutorial"], repr(c["text"])
    # statuses come back per sentence in a batch
    texts = [c["text"] for c in fw["sentences"]]
    ids, off, st = wp.encode_ids_batch(texts)
    want = np.array([1 if c["pretrained"] == "TIMEOUT" else (2 if c["pretrained"] == "INDEXERROR" else 0) for c in fw["sentences"]])
    assert np.array_equal(st, want)
    for i, c in enumerate(fw["sentences"]):
        if want[i]:
            assert off[i] == off[i + 1]
        else:
            assert wp._decode(ids[int(off[i]):int(off[i + 1])]) == c["pretrained"]


def test_wp_odd_vocabularies(swt, dev, golden):
    """tries with '##' corners, tokens containing spaces/punctuation, empty vocabulary"""
    for o in golden("fuzz_wp.json")["odd"]:
        tok = swt.FastWP()
        tok.vocab = set(o["vocab"])
        tok._build_trie()
        for c in o["cases"]:
            try:
                got = tok.tokenize(c["text"])
            except RuntimeError:
                got = "TIMEOUT"
            except IndexError:
                got = "INDEXERROR"
            assert got == c["tokens"], (o["vocab"], c["text"])


def test_wp_edge_shapes(wp, wp_orc):
    cases = [
        [],
        [""],
        ["", "", ""],
        ["", "a", "", "b", ""],
        ["słowo " * 3000],            # one sentence longer than a chunk (global-memory walker)
        ["x" * 20000],
        ["nie wiem " * 700, "a", "tak " * 1200, ""],
        ["a" * 4095, "b" * 4096, "c" * 4097, "d" * 2047, "e" * 2048, "f" * 2049],
        ["wyraz"] * 3000,
        ["hello!", "a ## b", "(a", "ok", "abc€def", "dobrze"],   # non-terminating ones mixed with fine ones
        ["5×2km", "˝zgoda˝", "áb", "zażółć gęślą jaźń"],
    ]
    for texts in cases:
        same_wp(wp, wp_orc, texts)


def test_wp_dedup_path_equals_direct_path(swt, wp, wp_orc, dev, golden, corpora, ref_dir):
    """word-level dedup inside a call (chunks between whitespace; default for batches >= 2.75 MiB when no vocabulary token
    holds whitespace) against the oracle and against the direct path, statuses included"""
    fw = golden("fuzz_wp.json")
    fuzz = [c["text"] for c in fw["sentences"]]
    cases = [
        [], [""], ["", "", ""], ["", "a", "", "b", ""], ["a"], ["a", "b", "a b", " a  b ", "\ta\nb\r"],
        ["słowo " * 3000], ["x" * 20000], ["nie wiem " * 700, "a", "tak " * 1200, ""],
        ["a" * 4095, "b" * 4096, "c" * 4097, "d" * 2047, "e" * 2048, "f" * 2049],
        ["wyraz"] * 3000, ["w " * 2500],                           # more words in a tile than the LDS path holds
        ["hello!", "a ## b", "(a", "ok", "abc€def", "dobrze"],     # non-terminating chunks: the whole sentence is refused
        ["dobrze hello! dobrze", "ok ok ok", "a ## b a", "tak (a tak"],
        ["5×2km", "˝zgoda˝", "áb", "zażółć gęślą jaźń", "a\u00a0b", "a\u2028b c\u3000d"],
        fuzz, corpora["pan"][:400],
    ]
    tut = swt.FastWP()
    tut.load_resources(os.path.join(ref_dir, "resources/tests/FastWordPiece"))
    from oracle import oracle as O
    tut_orc = O.OracleWP(tut._tokens)
    with dedup(dev, dev.DEDUP_ALWAYS, wp, tut):  # dedup whatever the size
        for texts in cases:
            same_wp(wp, wp_orc, texts)
            same_wp(tut, tut_orc, texts)
        got_d = wp.encode_ids_batch(corpora["t5k"])
    with dedup(dev, dev.DEDUP_NEVER, wp):
        got_n = wp.encode_ids_batch(corpora["t5k"])
    for a, b in zip(got_d, got_n):
        assert np.array_equal(a, b)
    for _ in range(3):  # the table is reused under a new epoch
        for a, b in zip(wp.encode_ids_batch(corpora["t5k"]), got_n):
            assert np.array_equal(a, b)


def test_wp_v30k_subsample_and_properties(swt, oracle, dev):
    """config 3 shape: V30k vocabulary, Zipf corpus; oracle parity on a 20,000-sentence subsample"""
    from subword_tokenizers_amd import synth

    vocab = synth.v30k()
    tok = swt.FastWP()
    tok.vocab = set(vocab)
    tok._build_trie()
    text, off = synth.wp_corpus(100000, seed=1000000, vocab=vocab)
    ids, ooff, st = tok._trie.encode(text, off)
    assert not st.any()  # the generator only uses characters that are single-char tokens: the reference terminates
    sub = synth.unpack(text, off, 0, 20000)
    orc = oracle.OracleWP(tok._tokens)
    oids, oooff, ost = orc.tokenize_batch_ids(sub)
    assert np.array_equal(ids[:int(ooff[20000])], oids) and np.array_equal(ooff[:20001], oooff)
    # batch independence: the same sentences in a smaller batch give the same ids
    t2, o2 = synth.wp_corpus(100000, seed=1000000, vocab=vocab)
    assert np.array_equal(t2, text)
    ids_b, off_b, _ = tok._trie.encode(text[:int(off[5000])].copy(), off[:5001].copy())
    assert np.array_equal(ids_b, ids[:int(ooff[5000])])
    # every emitted id is a vocabulary index or the UNK id
    assert int(ids.max()) <= len(vocab)


# ---------------------------------------------------------------------------------------------- BPE train

def test_train_tutorial_kat(swt, dev, golden):
    """reference resources/tests/FastBPE/merges.json: README tutorial corpus, max_vocab=25"""
    tok = swt.FastBPE()
    tok.train(["This is a sentence.", "Another example sentence."], 25)
    assert [list(p) for p in tok.merges_list] == golden("ref/resources/tests/FastBPE/merges.json")
    assert len(tok.vocab) == 25 and tok._bpe_ranks[("e", "n")] == 0
    assert tok.tokenize("This sentence") == ["this", "sentence"]


def test_train_micro_tie_breaks(swt, dev, golden):
    for c in golden("bpe_train_micro.json"):
        tok = swt.NaiveBPE()
        tok.train(list(c["corpus"]), c["max_vocab"])
        assert [list(p) for p in tok.merges_list] == c["merges"], (c["corpus"], c["max_vocab"])
        assert len(tok.vocab) == c["vocab_size"]


def test_train_5k_config1(swt, dev, golden, corpora):
    """config 1: train-5K, max_vocab=1000 -> the reference's 922 merges and its tokenization digest"""
    import hashlib
    import json

    g = golden("bpe_train5k_1000.json")
    tok = swt.FastBPE()
    tok.train(corpora["t5k"], 1000)
    assert [list(p) for p in tok.merges_list] == g["merges"]
    assert len(tok.vocab) == 1000
    toks = tok.tokenize_batch(corpora["t5k"])
    digest = hashlib.sha256(json.dumps(toks, ensure_ascii=False).encode("utf-8")).hexdigest()
    assert digest == g["tokens_sha256"]
    # corpus_as_symbols (bpe.py:23): the final segmentation of every unique word with its frequency
    cas = tok.corpus_as_symbols
    assert len(cas) == 22971 and sum(f for _, f in cas) == 80161
    # train() hands the device's own symbol ids to the rank table; a table built from the merges alone names them the same
    assert tok._train_ids is not None and len(tok._train_ids[0]) == 922
    tok2 = swt.FastBPE()
    tok2.merges_list = list(tok.merges_list)
    tok2._build_table()
    assert tok2._syms.strings == tok._syms.strings and tok2._bpe_ranks == tok._bpe_ranks
    a, ao = tok.encode_ids_batch(corpora["t5k"][:500])
    b, bo = tok2.encode_ids_batch(corpora["t5k"][:500])
    assert np.array_equal(a, b) and np.array_equal(ao, bo)


def test_train_state_matches_oracle_stepwise(swt, oracle, dev, corpora):
    """histogram and stream after every merge vs the oracle's full recount (exactness of the incremental update)"""
    sents = corpora["pan"][:300]
    text, off = dev.pack_utf8([s.lower() for s in sents])
    tr = dev.BpeTrainer.from_text(text, off)
    orc = oracle.OracleBPETrainer(sents)
    syms0, woff0, freq0 = orc.export()
    ds, dw, df = tr.export()
    assert np.array_equal(ds, syms0) and np.array_equal(dw, woff0) and np.array_equal(df, freq0)
    from subword_tokenizers_amd.tokenizers import _SymbolTable

    st = _SymbolTable()
    for step in range(60):
        left, right, count, tied, pos = tr.best()
        orc.run(10 ** 9, 1)
        ids, cnt = orc.merge_ids()
        assert (left, right, count) == (int(ids[-1][0]), int(ids[-1][1]), int(cnt[-1])), step
        merged = st.intern(st.string(left) + st.string(right))
        assert merged == int(ids[-1][2])
        tr.apply(left, right, merged)
        if step % 10 == 9:
            ds, dw, _ = tr.export()
            os_, ow, _ = orc.export()
            assert np.array_equal(ds, os_) and np.array_equal(dw, ow)
            # histogram == full recount of the oracle's stream
            keys, cnts = tr.histogram()
            want = {}
            for w in range(len(ow) - 1):
                seq = os_[int(ow[w]):int(ow[w + 1])]
                for a, b in zip(seq[:-1], seq[1:]):
                    k = (int(a) << 32) | int(b)
                    want[k] = want.get(k, 0) + int(freq0[w])
            got = {int(k): int(c) for k, c in zip(keys, cnts)}
            assert got == want


def test_train_device_driven_run_equals_stepwise(dev, oracle, corpora):
    """swt_bpe_train_run (K merges per host round trip) == best/apply one at a time == the oracle"""
    sents = corpora["t5k"][:1500]
    text, off = dev.pack_utf8([s.lower() for s in sents])
    a = dev.BpeTrainer.from_text(text, off)
    b = dev.BpeTrainer.from_text(text, off)
    orc = oracle.OracleBPETrainer(sents)
    orc.run(10 ** 9, 300)
    want, cnt = orc.merge_ids()
    la, ra, ca = a.run(300, 0x110000)
    assert len(la) == 300
    got = np.stack([la, ra, 0x110000 + np.arange(300, dtype=np.uint32)], axis=1)
    assert np.array_equal(got, want) and np.array_equal(ca, cnt)
    for i in range(300):
        l, r, c, tied, pos = b.best()
        assert (l, r, c) == (int(la[i]), int(ra[i]), int(ca[i]))
        b.apply(l, r, 0x110000 + i)
    sa, wa, _ = a.export()
    sb, wb, _ = b.export()
    so, wo, _ = orc.export()
    assert np.array_equal(sa, sb) and np.array_equal(wa, wb) and np.array_equal(sa, so) and np.array_equal(wa, wo)
    # run() past exhaustion returns fewer steps than asked
    t = dev.BpeTrainer.from_words(np.array([97, 98, 97, 98], dtype=np.uint32), np.array([0, 2, 4], dtype=np.uint64),
                                  np.array([1, 1], dtype=np.uint32))
    l, r, c = t.run(10, 0x110000)
    assert list(zip(l.tolist(), r.tolist(), c.tolist())) == [(97, 98, 2)]


def test_train_device_word_census_edge_shapes(swt, oracle, dev, corpora):
    """a1/a2 on the device (swt_words.hip): sentences longer than a chunk, words longer than a chunk, words of 255+ bytes
    (never deduplicated: same merges), punctuation runs, multi-byte text, empty sentences"""
    cases = [
        (["słowo " * 400 + "koniec", "", "x", "słowo słowo"], 60),
        (["ab" * 200 + " " + "ab" * 200 + " cd cd", "ab" * 200], 40),                 # 400-byte words, repeated
        (["q" * 3000 + " q q", "q" * 3000], 12),                                      # one word longer than a chunk, twice
        (["!!!???...,,, a,b;c", "(a)(b)(c)", "a-b-c-d " * 50], 30),
        (["zażółć gęślą jaźń " * 60, "ŻÓŁĆ żółć", "中文 中文 字 字"], 50),
        (corpora["pan"][:150] + ["", " ", "\t"], 200),
        # more than 64 sentences: the texts go to the device joined with U+0000 (swt_bpe_train_create_joined) ...
        (corpora["pan"][:120] + ["zażółć gęślą jaźń"] * 5, 150),
        # ... unless a sentence needs the host's str.lower() (U+0130, final sigma) or holds U+0000 itself
        (corpora["pan"][:120] + ["İstanbul ΣΑΣ ΌΣΟΣ"] * 3, 150),
        (corpora["pan"][:120] + ["nul \x00 inside", "\x00"], 150),
    ]
    for corpus, max_vocab in cases:
        tok = swt.NaiveBPE()
        tok.train(list(corpus), max_vocab)
        orc = oracle.OracleBPETrainer(corpus)
        orc.run(max_vocab)
        assert tok.merges_list == orc.merges_list, corpus[0][:40]
        assert len(tok.vocab) == orc.vocab_size
    # the unique-word stream itself (first-occurrence order, frequencies, symbols) on ordinary text
    sents = corpora["t5k"][:800]
    text, off = dev.pack_utf8([s.lower() for s in sents])
    tr = dev.BpeTrainer.from_text(text, off)
    ds, dw, df = tr.export()
    os_, ow, of = oracle.OracleBPETrainer(sents).export()
    assert np.array_equal(ds, os_) and np.array_equal(dw, ow) and np.array_equal(df, of)


def test_train_create_words_and_exhaustion(dev, oracle):
    """from an explicit word list; training to exhaustion stops with count == 0 (bpe.py:98-99)"""
    syms = np.array([ord(c) for c in "aaaabab"], dtype=np.uint32)
    woff = np.array([0, 4, 7], dtype=np.uint64)
    freq = np.array([3, 2], dtype=np.uint32)
    tr = dev.BpeTrainer.from_words(syms, woff, freq)
    assert tr.info()["n_base_symbols"] == 2
    seen = []
    nxt = 0x110000
    while True:
        l, r, c, tied, pos = tr.best()
        if c == 0:
            break
        seen.append((l, r, c))
        tr.apply(l, r, nxt)
        nxt += 1
    # 'aaaa'x3 + 'bab'x2: (a,a) counts 9 first; overlapping occurrences merge left to right
    assert seen[0] == (ord("a"), ord("a"), 9)
    ds, dw, _ = tr.export()
    assert list(np.diff(dw)) == [1, 1]


def test_train_config3_shape_words_with_frequencies(dev, oracle):
    """BASELINE configs[3] in shape (reference formulation: deduplicated word types with Zipf frequencies, scaled to 200 k types
    / 20 M tokens): the first 300 merges, pairs and counts, against the oracle's full recount"""
    from subword_tokenizers_amd import synth
    sym, off, freq = synth.train_words(200000, 1073741824, total_tokens=20_000_000)
    tr = dev.BpeTrainer.from_words(sym, off, freq)
    lefts, rights, counts = tr.run(300, dev.SYM_BASE)
    orc = oracle.OracleBPETrainer.from_words(sym, off, freq)
    orc.run(10 ** 9, 300)
    ids, cnt = orc.merge_ids()
    assert len(lefts) == 300 == len(ids)
    assert np.array_equal(np.asarray(lefts, dtype=np.uint32), ids[:, 0]) and np.array_equal(np.asarray(rights, dtype=np.uint32), ids[:, 1])
    assert np.array_equal(np.asarray(counts, dtype=np.uint64), cnt)
    got_sym, got_off, got_freq = tr.export()
    want_sym, want_off, want_freq = orc.export()
    assert np.array_equal(got_off, want_off) and np.array_equal(got_sym, want_sym) and np.array_equal(got_freq, want_freq)
    tr.close()


def test_train_plateaus_many_merges_per_step(dev, oracle):
    """The fast path merges SEVERAL tied pairs per step when they cannot affect each other (csrc/swt_bpe_train.hip,
    fast_apply_kernel); the reference takes them one at a time (bpe.py:88-111).  Small alphabets and flat frequencies make wide
    plateaus, shared symbols, twins and chains (the new pair of one merge tying with the rest): pairs, counts, their order and
    the final stream must be the oracle's, also when the merges are asked for in small, odd slices."""
    rng = np.random.default_rng(20260104)
    shapes = [(4, 3000, 2, 9, 1, 400), (8, 6000, 2, 12, 1, 700), (26, 20000, 1, 10, 1, 900), (6, 5000, 3, 7, 3, 500),
              (3, 800, 4, 16, 2, 300), (12, 50000, 2, 8, 1, 600)]
    for alpha, n_words, lo, hi, fmax, n_merges in shapes:
        lens = rng.integers(lo, hi + 1, size=n_words)
        off = np.zeros(n_words + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        sym = (97 + rng.integers(0, alpha, size=int(off[-1]))).astype(np.uint32)
        freq = rng.integers(1, fmax + 1, size=n_words).astype(np.uint32)
        orc = oracle.OracleBPETrainer.from_words(sym, off, freq)
        orc.run(10 ** 9, n_merges)
        ids, cnt = orc.merge_ids()
        for slices in (None, (1, 2, 7, 64, 5, 300)):
            tr = dev.BpeTrainer.from_words(sym, off, freq)
            ls, rs, cs = [], [], []
            i = 0
            while len(ls) < len(ids):
                ask = len(ids) - len(ls) if slices is None else min(slices[i % len(slices)], len(ids) - len(ls))
                l, r, c = tr.run(ask, dev.SYM_BASE + len(ls))
                assert len(l) == ask, (alpha, n_words, len(ls), ask, len(l))
                ls += l.tolist(); rs += r.tolist(); cs += c.tolist()
                i += 1
            got = np.stack([np.asarray(ls, dtype=np.uint32), np.asarray(rs, dtype=np.uint32)], axis=1)
            bad = np.nonzero((got != ids[:, :2]).any(axis=1))[0]
            assert bad.size == 0, (alpha, n_words, slices, int(bad[0]), got[bad[0]].tolist(), ids[bad[0]].tolist())
            assert np.array_equal(np.asarray(cs, dtype=np.uint64), cnt)
            gs, go, gf = tr.export()
            ws, wo, wf = orc.export()
            assert np.array_equal(go, wo) and np.array_equal(gs, ws)
            # the merges of one step log the same live-symbol count (the step's), and every step removes symbols
            carried = len(ids) / max(1, len(np.unique(tr.step_trace()[:, 3])))
            tr.close()
            assert carried > 1.2, "the steps carried %.2f merges each: the plateaus were not batched" % carried


def test_pack_and_lower_by_separator_and_by_code_points(dev, corpora):
    """_native.pack_and_lower: the U+0000-joined form (swt_utf8_prepare_joined) and the code-point form (swt_utf8_prepare, taken
    when a text holds U+0000 itself) both give the bytes and offsets of [t.lower() for t in texts] (utils.py:27)"""
    base = corpora["pan"][:400] + ["", "ŻÓŁĆ gęślą", "İstanbul ΣΑΣ ς", "a" * 5000, "", "\U0001F600 emoji", " "] + corpora["t5k"][:300]
    for texts in (base, base + ["nul \x00 inside", "\x00"], [""] * 100, ["x"] * 70, ["", "", "ß" * 9000] + [""] * 70):
        text, off = dev.pack_and_lower(list(texts))
        want = [t.lower().encode("utf-8", "surrogatepass") for t in texts]
        assert off.tolist() == np.concatenate([[0], np.cumsum([len(w) for w in want])]).tolist()
        assert text.tobytes() == b"".join(want)


def test_joined_text_with_too_few_separators_is_rejected(dev, swt, ref_dir):
    """include/swt.h: "SWT_ERR_INVALID when the separators do not add up".  The joined text holds 2 separators, the caller
    announces 2,000 sentences: the output buffer is 1,997 bytes SHORTER than what the split would write if it trusted the
    announcement (round 2 wrote past it: the mismatch only had to exceed the 64-byte pad).  Every entry point that takes the
    joined form must refuse, and the handle must stay usable."""
    import ctypes as C

    N = dev
    parts = [b"a" * 9000, b"b" * 9000, b"c" * 9000]
    joined = np.frombuffer(b"\x00".join(parts), dtype=np.uint8).copy()
    n_claim = 2000
    out = np.zeros(joined.size, dtype=np.uint8)
    off = np.zeros(n_claim + 1, dtype=np.uint64)
    need = np.zeros(n_claim, dtype=np.uint8)
    rc = N.lib().swt_utf8_prepare_joined(N.ptr(joined, N.u8p), int(joined.size), n_claim, N.ptr(out, N.u8p), N.ptr(off, N.u64p), N.ptr(need, N.u8p))
    assert rc == N.ERR_INVALID and b"separators" in N.lib().swt_last_error()
    bpe = swt.FastBPE()
    bpe.load_resources(os.path.join(ref_dir, "resources", "pretrained", "FastBPE"))
    with pytest.raises(N.SwtError) as e:
        bpe._ensure_table().encode_joined(joined, n_claim)
    assert e.value.code == N.ERR_INVALID
    h = C.c_void_p()
    rc = N.lib().swt_bpe_train_create_joined(N.ptr(joined, N.u8p), int(joined.size), n_claim, N.ptr(need, N.u8p), C.byref(h))
    assert rc == N.ERR_INVALID and not h.value
    # more separators than announced is refused as well, and a correct call still works afterwards
    many = np.frombuffer(b"\x00".join([b"x y"] * 500), dtype=np.uint8).copy()
    rc = N.lib().swt_utf8_prepare_joined(N.ptr(many, N.u8p), int(many.size), 100, N.ptr(out, N.u8p), N.ptr(off, N.u64p), N.ptr(need, N.u8p))
    assert rc == N.ERR_INVALID
    ids, ooff = bpe.encode_ids_batch(["Ala ma kota"] * 100)
    assert ooff.size == 101 and ids.size == 100 * (int(ooff[1]))


def _wp_order_cases(golden, ref_dir):
    import json
    for c in golden("wp_train_order.json"):
        if "corpus" in c:
            yield c, c["corpus"]
        else:
            with open(os.path.join(os.path.dirname(ref_dir), c["corpus_ref"]["file"]), encoding="utf-8") as f:
                yield c, json.load(f)[: c["corpus_ref"]["first"]]


def test_wp_train_matches_reference_merge_order(swt, dev, golden, ref_dir):
    """NaiveWP.train on the device -- initial symbols, the sequence of merges and the final vocabulary of the
    reference (wordpiece.py:29-103) on micro corpora (twins, ties, exhaustion), pan_tadeusz and train-5K"""
    n = 0
    for c, corpus in _wp_order_cases(golden, ref_dir):
        m = swt.NaiveWP()
        m.train(list(corpus), 0)
        assert sorted(m.vocab) == c["initial"]
        m.train(list(corpus), c["max_vocab"])
        assert [list(p) for p in m._merge_order] == c["merges"], (c.get("corpus", c.get("corpus_ref")), m._merge_order[:5])
        assert sorted(m.vocab) == c["vocab"]
        n += len(c["merges"])
    assert n > 300
    for c in golden("wp_train_micro.json"):
        m = swt.NaiveWP()
        m.train(list(c["corpus"]), c["max_vocab"])
        assert sorted(m.vocab) == c["vocab"], c["corpus"]


def test_wp_train_state_matches_oracle(swt, oracle, dev, corpora):
    """device stream (symbols, frequencies) after every few merges against the oracle's full recount; FastWP.train end to end"""
    corpus = corpora["t5k"][:1500]
    m = swt.NaiveWP()
    orc = oracle.OracleWPTrainer(corpus)
    base = orc.vocab_size
    for extra in (1, 7, 40, 200):
        m.train(list(corpus), base + extra)
        o = oracle.OracleWPTrainer(corpus)
        o.run(base + extra)
        assert [list(p) for p in m._merge_order] == [list(p) for p in o.merges_list]
        want = set(o.merged_tokens)
        assert len(m.vocab) == o.vocab_size and want <= m.vocab
        syms, woff, freq = o.export()
        got = m.corpus_as_symbols
        assert len(got) == len(woff) - 1
        for w in (0, 1, 2, len(got) // 2, len(got) - 1):
            assert got[w][0] == [o.symbol(int(x)) for x in syms[int(woff[w]):int(woff[w + 1])]] and got[w][1] == int(freq[w])
    fw = swt.FastWP()
    fw.train(list(corpus), base + 200)
    assert fw.vocab == m.vocab
    assert fw.tokenize("Ala ma kota") == fw.tokenize("ala ma kota")


@pytest.fixture(params=["fast", "generic"])
def shard_mode(request, monkeypatch):
    """the two forms of the sharded runner: the fast two-launch step with several tied merges per step (default), and round 2's
    one-merge-per-step form (SWT_DIST_GENERIC=1: the fallback while a plateau is wider than the candidate list)"""
    monkeypatch.setenv("SWT_DIST_GENERIC", "1" if request.param == "generic" else "0")
    return request.param


def test_sharded_training_loopback_runner(swt, oracle, dev, corpora, shard_mode):
    """csrc/swt_dist.hip through the loop-back communicator: 2, 3 and 5 shards as trainers of this process on one GPU -- local
    histograms reduced once, per step one tie-message gather + one record-block gather, ties broken by first occurrence over
    (rank, word, offset) -- must reproduce the single-shard merges exactly; the tiny-alphabet corpus makes almost every step a tie"""
    from subword_tokenizers_amd.distributed import train_sharded_loopback

    sents = corpora["pan"][:400]
    ref = oracle.OracleBPETrainer(sents)
    ref.run(400)
    want = [tuple(p) for p in ref.merges_list]
    for world in (2, 3, 5):
        merges, stats = train_sharded_loopback(sents, 400, world)
        assert merges == want, world
        assert all(not (st["flags"] & 1) for st in stats)  # the inverted index stayed in use on every shard
    ties = ["ab ba ab", "ba ab cc", "cc ab ba", "abab baba", "cab bac", "ccc aaa bbb", "abc cba", "bb aa"]
    ref = oracle.OracleBPETrainer(ties)
    ref.run(20)
    for world in (2, 4):
        merges, _ = train_sharded_loopback(ties, 20, world)
        assert merges == [tuple(p) for p in ref.merges_list], world
    # more shards than sentences: empty shards take part in every collective
    merges, _ = train_sharded_loopback(ties[:2], 12, 4)
    ref = oracle.OracleBPETrainer(ties[:2])
    ref.run(12)
    assert merges == [tuple(p) for p in ref.merges_list]


def test_sharded_training_block_overflow_recovers(swt, oracle, dev, corpora, monkeypatch, shard_mode):
    """a merge whose deltas do not fit the record block halts the batch on every shard; the runner grows the blocks and repeats
    that exchange -- t5k's first merges touch hundreds of pairs; the block starts at 64 records here (4,096 by default)"""
    from subword_tokenizers_amd import _native as N

    monkeypatch.setenv("SWT_DIST_BLOCK_RECORDS", "64")
    from subword_tokenizers_amd.distributed import HipShardEngine, ShardedBpeTrainer

    sents = corpora["t5k"][:3000]
    ref = oracle.OracleBPETrainer(sents)
    target = ref.vocab_size + 300
    ref.run(target)
    comm = N.Dist.loopback(2)
    trainers = []
    for r in range(2):
        text, off = N.pack_and_lower(ShardedBpeTrainer.shard(sents, r, 2))
        trainers.append(N.BpeTrainer.from_text(text, off))
    tr = ShardedBpeTrainer(HipShardEngine(trainers, comm), 0, 2)
    merges = tr.train(target)
    assert [tuple(m) for m in merges] == [tuple(p) for p in ref.merges_list]
    if shard_mode == "generic":
        assert trainers[0].stats()["steps"] > 300  # halted steps were spent: the overflow path ran
    tr.engine.close()
    comm.close()


def test_sharded_training_over_rccl_world1(swt, oracle, dev, corpora, shard_mode):
    """the RCCL side of csrc/swt_dist.hip on the one GPU there is: a communicator of ONE rank (ncclGetUniqueId,
    ncclCommInitRank, the per-merge ncclAllGather pair on the training stream) must train exactly like the unsharded path"""
    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd.distributed import ShardedBpeTrainer

    sents = corpora["t5k"][:1200]
    ref = oracle.OracleBPETrainer(sents)
    target = ref.vocab_size + 250
    ref.run(target)
    comm = N.Dist.rccl(0, 1, N.Dist.unique_id())
    tr = ShardedBpeTrainer.from_corpus(sents, 0, 1, comm)
    try:
        merges = tr.train(target)
        assert [tuple(m) for m in merges] == [tuple(p) for p in ref.merges_list]
        assert len(tr.vocab) == ref.vocab_size
    finally:
        tr.engine.close()
        comm.close()


@pytest.mark.gpu
def test_bpe_word_lane_kernel_boundaries(swt, oracle, dev, bpe, bpe_orc):
    """the shapes bpe_lane_kernel (csrc/swt_bpe_encode.hip) has of its own: words around the 32-slot live mask (longer ones take
    slow_word), more multi-symbol words in a chunk than a wave has lanes (the lanes take a next word), words and multi-byte
    characters straddling the 64-byte blocks of the split and the 512-byte chunks, pairs that recur inside a word, and the same
    texts under a PROPER table (one occurrence of the best pair per round) and under tables that are not (every word through
    slow_word: all occurrences per round, bpe.py:221-235)."""
    def table(merges):
        tok = swt.FastBPE()
        tok.merges_list = list(merges)
        tok._build_table()
        return tok, oracle.OracleBPE(merges)

    # proper: a chain that merges long words in many rounds, with recurring pairs and a twin
    proper = [("a", "b"), ("c", "d"), ("ab", "cd"), ("e", "e"), ("ee", "e"), ("abcd", "ab"), ("x", "y"), ("xy", "xy"),
              ("ab", "ab"), ("d", "a"), ("ż", "ó"), ("żó", "ł"), ("b", "c"), ("abcdab", "cd"), ("y", "x")]
    # not proper: (ab, c) ranks BELOW the merge that makes ab, and (aa, a) below (a, a)
    improper = [("ab", "c"), ("a", "b"), ("aa", "a"), ("a", "a"), ("c", "d"), ("x", "y"), ("b", "c")]
    texts = []
    for n in (30, 31, 32, 33, 34, 63, 64, 65, 100):                       # symbols per word around the live mask
        texts.append("abcd" * (n // 4) + "abcd"[: n % 4] + " " + "e" * n + " " + "xy" * (n // 2))
        texts.append(("żół" * n)[:n] + " ab " + "abab" * (n // 4))
    texts.append(" ".join("ab cd abcd xyxy eee abab".split() * 40))      # ~240 multi-symbol words in one sentence: refill
    texts.append(" ".join(["abcdabcd" * 3, "eeeeeeeeee", "xyxyxyxy"] * 30 + ["ab"] * 200))
    for shift in range(0, 70, 7):                                        # word / multi-byte character across block and chunk ends
        texts.append("q" * shift + " " + "abcdabcdab" * 6 + " żółżółżół " + "abab" * 3)
        texts.append("ż" * (250 + shift) + " " + "abcd" * 70 + " e" * 40)
        texts.append(("ab " * 170)[: 500 + shift] + "abcdabcd" * 5)
    texts += ["ab", "a b", "ab.cd,ab!", "", " ", "abcdabcdabcdabcdabcdabcdabcdabcdabcdab" * 9]
    for merges in (proper, improper):
        tok, orc = table(merges)
        same_bpe(tok, orc, texts)
        for t in texts[:12]:
            same_bpe(tok, orc, [t])
        with dedup(dev, dev.DEDUP_ALWAYS, tok):
            same_bpe(tok, orc, texts)
    lib = dev.lib()
    import ctypes as C
    lib.swt_debug_bpe_table_info.restype = C.c_int
    lib.swt_debug_bpe_table_info.argtypes = [C.c_void_p, C.c_int]
    tp, _ = table(proper)
    ti, _ = table(improper)
    assert lib.swt_debug_bpe_table_info(tp._table._h, 2) == 1 and lib.swt_debug_bpe_table_info(ti._table._h, 2) == 0
    same_bpe(bpe, bpe_orc, texts)  # and the pretrained table
