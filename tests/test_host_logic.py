"""CPU-side checks of the product: the C ABI exports what include/swt.h declares, the host-side classes mirror
the reference's surface and error behaviour, the C++ trie build matches the reference's WPTrie_E2E, and compute
calls fail loudly without a GPU (there is no CPU fallback).  No kernel is launched here."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "swt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(swt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(native):
    names = declared_functions()
    assert len(names) >= 30
    lib = ctypes.CDLL(native._build.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libswt_hip.so does not export %s" % n
    # and the ctypes binding covers exactly the header
    assert sorted(native.SIGNATURES) == names


def test_abi_exception_barrier(native):
    """include/swt.h: "nothing throws, aborts".  Every extern "C" entry point is a function-try-block (SWT_API_CATCH): a C++
    exception raised inside the library comes back as a status + swt_last_error(), never as std::terminate -> SIGABRT (the
    suspected shape of round 2's abort in swt_bpe_train_run).  swt_abi_selftest throws on purpose, behind the same barrier."""
    lib = native.lib()
    assert lib.swt_abi_selftest(0) == 0
    for kind, code, word in ((1, native.ERR_NOMEM, "bad_alloc"), (2, native.ERR_NOMEM, "length_error"),
                             (3, native.ERR_INTERNAL, "swt_abi_selftest"), (4, native.ERR_INTERNAL, "unknown")):
        assert lib.swt_abi_selftest(kind) == code
        assert word in lib.swt_last_error().decode()
    with pytest.raises(native.SwtError):
        native.check(lib.swt_abi_selftest(3))


def test_every_entry_point_has_the_exception_barrier():
    """source check: each function include/swt.h declares is defined as `... swt_x(...) try {` (or is one of the two
    accessors that cannot throw), so none can be added without its barrier"""
    srcs = ""
    cs = os.path.join(ROOT, "subword-tokenizers_amd", "csrc")
    for f in os.listdir(cs):
        if f.endswith(".hip"):
            srcs += open(os.path.join(cs, f), encoding="utf-8").read()
    for name in declared_functions():
        if name in ("swt_last_error", "swt_version", "swt_unidata_version"):
            continue
        m = re.search(r"^[a-z_0-9 \*]+\b%s\((?:[^;{]|\n)*?\)\s*(try)?\s*\{" % name, srcs, flags=re.M)
        assert m, "no definition of %s found" % name
        assert m.group(1) == "try", "%s is defined without the exception barrier" % name


def test_no_gpu_means_loud_failure(native, swt, ref_dir):
    if native.device_count() > 0:
        pytest.skip("a GPU is present")
    bpe = swt.FastBPE()
    bpe.load_resources(os.path.join(ref_dir, "resources/pretrained/FastBPE"))
    assert len(bpe.merges_list) == 19876 and len(bpe._bpe_ranks) == 19876
    with pytest.raises(swt.NoDeviceError):
        bpe.tokenize("ala ma kota")
    with pytest.raises(swt.NoDeviceError):
        bpe.train(["ala ma kota"], 30)
    wp = swt.FastWP()
    wp.load_resources(os.path.join(ref_dir, "resources/pretrained/FastWordPiece"))
    with pytest.raises(swt.NoDeviceError):
        wp.tokenize("ala ma kota")


def test_product_never_imports_the_oracle():
    """the product path must not import, link, load or execute anything under oracle/"""
    pkg = os.path.join(ROOT, "subword-tokenizers_amd")
    bad = re.compile(r"(import\s+oracle|from\s+oracle|liboracle|swt_oracle|oracle/|orc_[a-z_]+\()")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not bad.search(src), os.path.join(dirpath, f)


def test_lower_table_is_pythons_str_lower(native, golden):
    """the device lowercase table against str.lower() of this interpreter, every code point; the 26 it leaves to
    the host are exactly those whose lowercase changes length, expands, or depends on context (U+03A3)"""
    fx = golden("unicode_lower.json")
    host = set(fx["host"])
    assert len(host) == 26 and 0x3A3 in host and 0x130 in host
    n_pairs = 0
    for cp in range(0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            assert native.lower_of(cp) == cp
            continue
        low = chr(cp).lower()
        got = native.lower_of(cp)
        if cp in host:
            assert got == 0xFFFFFFFF
            assert cp == 0x3A3 or len(low) != 1 or len(low.encode()) != len(chr(cp).encode())
        else:
            assert low == chr(got), hex(cp)
            n_pairs += got != cp
    assert n_pairs == len(fx["pairs"]) == fx["meta"]["pairs"]


def test_class_table_matches_fixture(native, golden):
    fx = golden("unicode_classes.json")
    for bit, name in ((1, "bert_ws"), (2, "bert_punct"), (4, "py_space"), (8, "py_alnum")):
        for lo, hi in fx[name]:
            for cp in {lo, hi, (lo + hi) // 2}:
                assert native.class_of(cp) & bit, (name, hex(cp))
            for cp in (lo - 1, hi + 1):
                inside = any(a <= cp <= b for a, b in fx[name])
                if 0 <= cp < 0x110000 and not inside:
                    assert not native.class_of(cp) & bit, (name, hex(cp))


def test_preprocessing_matches_reference(swt, golden):
    base = swt.SubwordTokenizer()
    for case in golden("pretok_fuzz.json"):
        got = base.preprocessing([case["text"]])[0]
        assert [w for w, _ in got] == case["words"], repr(case["text"])
        low = case["text"].lower()
        for w, (a, b) in got:
            assert low[a:b] == w


def test_preprocessing_agrees_with_installed_wheel(swt):
    tokenizers = pytest.importorskip("tokenizers")
    pt = tokenizers.pre_tokenizers.BertPreTokenizer()
    base = swt.SubwordTokenizer()
    for s in ["Hello, World!", "A_b $5+3^2", "a\xa0b c", "x—y «z» 1,5%", "İstanbul abc€def", "  \t\n"]:
        assert base.preprocessing([s])[0] == pt.pre_tokenize_str(s.lower())


def test_constructor_accepts_reference_style_tokenizer(swt):
    tokenizers = pytest.importorskip("tokenizers")

    class Shim:
        pass

    hf = Shim()
    hf.backend_tokenizer = Shim()
    hf.backend_tokenizer.pre_tokenizer = tokenizers.pre_tokenizers.BertPreTokenizer()
    tok = swt.FastBPE(hf)
    assert tok.tokenizer is hf
    hf.backend_tokenizer.pre_tokenizer = tokenizers.pre_tokenizers.Whitespace()
    with pytest.raises(ValueError):
        swt.FastBPE(hf)


def test_type_errors_like_the_reference(swt):
    for cls in (swt.NaiveBPE, swt.FastBPE, swt.NaiveWP, swt.FastWP):
        tok = cls()
        with pytest.raises(TypeError):
            tok.train("not a list", 10)
        with pytest.raises(TypeError):
            tok.train(["a", 1], 10)
        with pytest.raises(TypeError):
            tok.train(["a"], "10")
        with pytest.raises(TypeError):
            tok.tokenize(["a"])


def test_resource_formats_round_trip(swt, tmp_path, ref_dir):
    bpe = swt.FastBPE()
    bpe.load_resources(str(tmp_path / "missing"))  # silently a no-op (bpe.py:187)
    assert bpe.merges_list == []
    bpe.load_resources(os.path.join(ref_dir, "resources/tests/FastBPE"))
    assert bpe.merges_list[0] == ("e", "n") and len(bpe.merges_list) == 10
    assert bpe._bpe_ranks[("t", "h")] == 1
    bpe.save_resources(str(tmp_path / "out" / "FastBPE"))
    saved = json.load(open(tmp_path / "out" / "FastBPE" / "merges.json", encoding="utf-8"))
    assert saved == json.load(open(os.path.join(ref_dir, "resources/tests/FastBPE/merges.json"), encoding="utf-8"))
    wp = swt.FastWP()
    wp.load_resources(os.path.join(ref_dir, "resources/tests/FastWordPiece"))
    wp.save_resources(str(tmp_path / "out" / "FastWordPiece"))
    assert set(json.load(open(tmp_path / "out" / "FastWordPiece" / "vocab.json", encoding="utf-8"))) == wp.vocab
    bpe.reset()
    assert bpe.merges_list == [] and bpe.vocab == set()


def test_duplicate_pairs_last_rank_wins(swt, tmp_path):
    d = tmp_path / "FastBPE"
    d.mkdir()
    json.dump([["a", "b"], ["c", "d"], ["a", "b"]], open(d / "merges.json", "w"))
    bpe = swt.FastBPE()
    bpe.load_resources(str(d))
    assert bpe._bpe_ranks == {("a", "b"): 2, ("c", "d"): 1}  # bpe.py:257


def test_naive_bpe_encode_is_the_didactic_loop(swt, golden, ref_dir):
    fz = golden("fuzz_bpe.json")
    nb = swt.NaiveBPE()
    nb.merges_list = [tuple(m) for m in golden("bpe_train5k_1000.json")["merges"]]
    # NaiveBPE == FastBPE on ordinary text (the author golden shows the same); compare on a few sentences
    for c in fz["sentences"][:12]:
        assert nb.tokenize(c["text"]) == c["t5k"], repr(c["text"])
    assert nb.encode_word("") == [] and nb._replace_pair(("a", "a"), list("aaa")) == ["aa", "a"]


def test_naive_wp_encode_and_train_needs_the_device(swt, native):
    if native.device_count() < 1:  # NaiveWP.train runs on the device (tests/test_gpu_parity.py covers it): no CPU fallback
        with pytest.raises(swt.NoDeviceError):
            swt.NaiveWP().train(["ala ma kota"], 30)
    with pytest.raises(TypeError):
        swt.NaiveWP().train("not a list", 30)
    m = swt.NaiveWP()
    m.vocab = {"un", "##aff", "##able", "a", "##b"}
    assert m.encode_word("unaffable") == ["un", "##aff", "##able"]
    assert m.encode_word("xyz") == ["[UNK]"]
    assert m.tokenize("Unaffable ab") == ["un", "##aff", "##able", "a", "##b"]


def test_trie_build_matches_reference_tutorial(swt, native, golden):
    """a9/a10: WPTrie_E2E insert + precompute (utils.py:75-139) restated in C++ (swt_wp.hip), full node dump"""
    g = golden("trie_digest.json")["tutorial"]
    tokens = sorted(g["vocab"])
    trie = native.WpTrie(tokens)
    assert trie.stats()["nodes"] == len(g["nodes"]) + 2  # + root, root_p
    ident = {0: "<ROOT>", 1: "<ROOT_P>"}
    for chars_seen, is_end, link_s, pops in g["nodes"]:
        info = trie.node(chars_seen)
        assert info is not None, chars_seen
        nid, link, end, pop_ids = info
        assert end == bool(is_end), chars_seen
        got_link = "<NONE>" if link < 0 else ident.get(link, trie.node_path(link))
        assert got_link == link_s, (chars_seen, got_link, link_s)
        assert [tokens[int(t)] for t in pop_ids] == pops, chars_seen


def test_trie_build_matches_reference_pretrained(native, golden):
    import hashlib

    g = golden("trie_digest.json")["pretrained"]
    vocab = golden("ref/resources/pretrained/FastWordPiece/vocab.json")
    tokens = sorted(vocab)
    trie = native.WpTrie(tokens)
    st = trie.stats()
    assert st["nodes"] == g["n_nodes"] + 2  # + root, root_p
    assert st["edges"] == g["n_nodes"]
    ident = {0: "<ROOT>", 1: "<ROOT_P>"}
    for chars_seen, is_end, link_s, pops in g["sample"]:
        nid, link, end, pop_ids = trie.node(chars_seen)
        assert end == bool(is_end)
        assert ("<NONE>" if link < 0 else ident.get(link, trie.node_path(link))) == link_s
        assert [tokens[int(t)] for t in pop_ids] == pops
    # full structure digest: rebuild the reference's dump order (sorted by chars_seen) from the C++ trie
    lines = []
    seen = set()  # every node = every prefix of a vocabulary entry (and of "##")
    for tok in tokens + ["##"]:
        for k in range(1, len(tok) + 1):
            seen.add(tok[:k])
    for path in sorted(seen):
        nid, link, end, pop_ids = trie.node(path)
        link_s = "<NONE>" if link < 0 else ident.get(link, trie.node_path(link))
        lines.append([path, int(end), link_s, [tokens[int(t)] for t in pop_ids]])
    assert len(lines) == g["n_nodes"]
    digest = hashlib.sha256(json.dumps(lines, ensure_ascii=False).encode("utf-8")).hexdigest()
    assert digest == g["sha256"]


def test_sharp_corner_evaluation(native, golden):
    """NaiveWP.encode_word("##") (wordpiece.py:260-261) is evaluated once at trie build"""
    cases = [
        (["a", "##a", "#", "##"], ["##"]),
        (["a", "##a"], ["[UNK]"]),
        (["a", "##a", "#"], None),  # '#' in vocab, '##' not: the reference never returns
        (["a", "##a", "#", "###"], ["#", "###"]),
        (["a", "##a", "#", "####"], ["#", "#", "####"]),
    ]
    for vocab, want in cases:
        tokens = sorted(vocab)
        trie = native.WpTrie(tokens)
        c = trie.corner()
        if want is None:
            assert c is None
        else:
            n = len(tokens)
            got = [tokens[int(t)] if t < n else "[UNK]" for t in c]
            assert got == want, vocab
    # the pretrained vocabulary has '#' but not '##' 
    pre = sorted(golden("ref/resources/pretrained/FastWordPiece/vocab.json"))
    assert native.WpTrie(pre).corner() is None


def test_trie_view_surface(swt, ref_dir):
    wp = swt.FastWP()
    wp.load_resources(os.path.join(ref_dir, "resources/tests/FastWordPiece"))
    t = wp.vocab_trie
    assert t.root.chars_seen == "" and t.root_sharp.chars_seen == "##" and t.root_p.failure_link is None
    node = t.root.child("t")
    assert node is not None and node.chars_seen == "t"
    assert t.root.child("中") is None
    assert {t.root, t.root_sharp, t.root_p} == {t.root, t.root_sharp, t.root_p}


def test_pack_helpers(native):
    buf, off = native.pack_utf8(["ab", "", "ż\U0001F600", "\ud800"])
    assert off.tolist() == [0, 2, 2, 8, 11] and buf.size == 11
    blob, off = native.pack_utf32(["ab", "", "ż"])
    assert off.tolist() == [0, 2, 2, 3] and blob.tolist() == [97, 98, 0x17C]


def test_join_texts_is_join_plus_encode(native):
    """csrc/swt_pyhost.c (one pass over the strings' PEP 393 data) against "\\0".join(texts).encode("utf-8", "surrogatepass"),
    every string kind, lone surrogates, U+0000 inside texts, and the Python form it stands in for"""
    import random

    def want(ts):
        return "\x00".join(ts).encode("utf-8", "surrogatepass"), sum(t.count("\x00") for t in ts)

    assert native.pyhost() is not None, "swt_pyhost.c did not build here (gcc and Python.h are part of the image)"
    rng = random.Random(11)
    alphabets = ["abc xyz\x00żłé中\U0001F600\ud800ÿ\x7f\x80߿ࠀ￿\U00010000", "abcdefgh ijklmnoł", "abc", "é\xff a", "\U0001F600a"]
    cases = [[], [""], ["", ""], ["a"], ["\x00"], ["ł" + "a" * 7, "a" * 7 + "ł", "a" * 8 + "ł" + "\x00" * 9, "ł" * 8, "中" * 17, "a" * 1000]]
    for trial in range(600):
        a = alphabets[trial % len(alphabets)]
        cases.append(["".join(rng.choice(a) for _ in range(rng.randrange(0, 70))) for _ in range(rng.randrange(0, 40))])
    for ts in cases:
        joined, n_nul = native.join_texts(ts)
        data, nul = want(ts)
        assert joined.tobytes() == data and n_nul == nul, ts
    # lists long enough for the threaded form (>= 4,096 strings and 1 MiB): every kind of string at the seams of the ranges,
    # with 1, 3 and 8 threads, and empty strings in bulk
    big = ["".join(rng.choice(alphabets[i % len(alphabets)]) for _ in range(rng.randrange(150, 400))) for i in range(6000)]
    big[0] = ""
    big[2999] = "\x00"
    big[-1] = "ł" * 4097 + "中"
    for threads in ("1", "3", "8"):
        monkey_env = os.environ.get("SWT_JOIN_THREADS")
        os.environ["SWT_JOIN_THREADS"] = threads
        try:
            joined, n_nul = native.join_texts(big)
        finally:
            if monkey_env is None:
                del os.environ["SWT_JOIN_THREADS"]
            else:
                os.environ["SWT_JOIN_THREADS"] = monkey_env
        data, nul = want(big)
        assert joined.tobytes() == data and n_nul == nul, threads
    # the seams: every string two-byte (the vector form stores 16-byte lanes) and short, so that a store of one range would
    # reach into the next range if it were allowed to (it was, once: a stray zero byte per seam on the GPU box)
    seam = [("ł" + "ab" * (i % 5)) for i in range(4096)] + ["x" * (1 << 20)]
    for threads in ("2", "8", "16"):
        os.environ["SWT_JOIN_THREADS"] = threads
        try:
            for _ in range(20):
                joined, n_nul = native.join_texts(seam)
                assert n_nul == 0 and int((joined == 0).sum()) == len(seam) - 1, threads
        finally:
            del os.environ["SWT_JOIN_THREADS"]
    assert joined.tobytes() == want(seam)[0]
    joined, n_nul = native.join_texts([""] * 5000 + ["a" * (1 << 20)] + [""] * 5000)
    assert joined.size == 10000 + (1 << 20) and n_nul == 0 and joined[5000:5000 + (1 << 20)].tobytes() == b"a" * (1 << 20)
    saved = native._pyhost
    try:
        native._pyhost = False  # the str.join + str.encode form
        for ts in cases[:60]:
            joined, n_nul = native.join_texts(ts)
            data, nul = want(ts)
            assert joined.tobytes() == data and n_nul == nul, ts
        with pytest.raises(TypeError, match="Text must be a string"):
            native.join_texts(["a", 3])
    finally:
        native._pyhost = saved
    with pytest.raises(TypeError, match="Text must be a string"):
        native.join_texts(["a", b"b", "c"])
    with pytest.raises(TypeError, match="corpus"):
        native.join_texts(("a", "b"), "corpus")

    class S(str):
        pass

    assert native.join_texts([S("ab"), "c"])[0].tobytes() == b"ab\x00c"


def test_nested_lists_both_forms(native):
    """ids -> List[List[str]] (the reference's output shape): csrc/swt_pyhost.c against the numpy form and a plain comprehension"""
    rng = np.random.default_rng(3)
    table = ["t%d" % i for i in range(300)]
    inv = rng.integers(0, 300, size=5000).astype(np.int32)
    cuts = np.sort(rng.integers(0, 5001, size=400))
    off = np.concatenate([[0], cuts, [5000]]).astype(np.uint64)
    want = [[table[int(t)] for t in inv[int(off[s]):int(off[s + 1])]] for s in range(off.size - 1)]
    assert native.nested_lists(table, inv, off) == want
    saved = native._pyhost
    try:
        native._pyhost = False
        assert native.nested_lists(table, inv, off) == want
        with pytest.raises(IndexError):
            native.nested_lists(table, np.array([300], dtype=np.int32), np.array([0, 1], dtype=np.uint64))
    finally:
        native._pyhost = saved
    with pytest.raises(IndexError):
        native.nested_lists(table, np.array([1, 300], dtype=np.int32), np.array([0, 2], dtype=np.uint64))
    with pytest.raises(IndexError):
        native.nested_lists(table, np.array([-1], dtype=np.int32), np.array([0, 1], dtype=np.uint64))
    with pytest.raises(ValueError):
        native.nested_lists(table, np.array([1, 2], dtype=np.int32), np.array([0, 3], dtype=np.uint64))
    assert native.nested_lists(table, np.zeros(0, np.int32), np.zeros(1, np.uint64)) == []
    assert native.nested_lists(table, np.array([5], np.int32), np.array([0, 0, 1, 1], np.uint64)) == [[], ["t5"], []]


def test_nested_lists_by_key_both_forms(native):
    rng = np.random.default_rng(4)
    key = rng.choice(np.array([3, 7, 2_000_001, 12, 999_999, 0], dtype=np.uint32), size=3000)
    off = np.concatenate([[0], np.sort(rng.integers(0, 3001, size=50)), [3000]]).astype(np.uint64)
    calls = []
    spell = lambda k: (calls.append(k), "k%d" % k)[1]
    want = [["k%d" % int(t) for t in key[int(off[s]):int(off[s + 1])]] for s in range(off.size - 1)]
    assert native.nested_lists_by_key(key, 2_000_002, spell, off) == want
    assert sorted(calls) == sorted(set(key.tolist())) and calls[0] == int(key[0])  # once per distinct key, first seen first
    saved = native._pyhost
    try:
        native._pyhost = False
        assert native.nested_lists_by_key(key, 2_000_002, spell, off) == want
        with pytest.raises(IndexError):
            native.nested_lists_by_key(key, 2_000_001, spell, off)
    finally:
        native._pyhost = saved
    with pytest.raises(IndexError):
        native.nested_lists_by_key(key, 2_000_001, spell, off)
    assert native.nested_lists_by_key(np.zeros(0, np.uint32), 10, spell, np.zeros(1, np.uint64)) == []


def test_unicode_database_mismatch_lowercases_on_the_host(native, monkeypatch):
    """the device's lowercase / class tables carry the Unicode version they were generated from; an interpreter on another
    version lowercases every batch itself (pack_and_lower needs no device then), whatever its size"""
    import unicodedata

    assert native.lib().swt_unidata_version().decode() == unicodedata.unidata_version  # this image: generated here
    assert native.device_lower_ok()
    monkeypatch.setattr(native, "_lower_ok", False)
    texts = ["Zażółć GĘŚLĄ jaźń %d" % i for i in range(100)] + ["İstanbul ΣΟΦΟΣ", ""]
    buf, off = native.pack_and_lower(texts)  # no GPU in this test: the device path would raise NoDeviceError
    want = [t.lower().encode("utf-8") for t in texts]
    assert buf.tobytes() == b"".join(want)
    assert off.tolist() == np.concatenate([[0], np.cumsum([len(w) for w in want])]).tolist()
    assert native.BpeTrainer.from_texts(texts) is None


def test_bench_watchdog_emits_the_line_without_the_train_block(monkeypatch):
    """bench.py at world > 1: if the sharded training block raises (or hangs) on a rank, rank 0 still writes the JSON line, with
    train = {"error": ...}, and every rank leaves without entering another collective -- with a NON-ZERO exit code: the
    driver must see a failed training leg as a failed run"""
    import time

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench

    class Left(BaseException):
        pass

    emitted = []
    monkeypatch.setattr(bench, "EMIT", emitted.append)
    monkeypatch.setattr(bench.os, "_exit", lambda code: (_ for _ in ()).throw(Left(code)))
    out = {"metric": "m", "value": 1.0}

    def boom(*a, **k):
        raise SystemExit("PARITY FAILURE: made up")

    monkeypatch.setattr(bench, "train_bench", boom)
    with pytest.raises(Left) as left:
        bench.train_bench_guarded(out, None, None, None, 0, 2, None, [], 8000, "x")
    assert left.value.args[0] != 0
    assert len(emitted) == 1 and "PARITY FAILURE" in emitted[0]["train"]["error"] and emitted[0]["value"] == 1.0
    emitted.clear()
    with pytest.raises(Left):  # another rank: nothing to print, it just leaves
        bench.train_bench_guarded(out, None, None, None, 1, 2, None, [], 8000, "x")
    assert emitted == []
    # a block that hangs: the timer fires in its own thread, writes the line and leaves (here: raises inside that thread)
    fired = []
    monkeypatch.setattr(bench.os, "_exit", lambda code: fired.append(code))
    monkeypatch.setattr(bench, "train_bench", lambda *a, **k: time.sleep(0.6) or {"ok": True})
    got = bench.train_bench_guarded(out, None, None, None, 0, 2, None, [], 8000, "x", limit_s=0.2)
    assert fired == [3] and len(emitted) == 1 and "not finished" in emitted[0]["train"]["error"]
    # and the ordinary case: the block's result, no line, nobody leaves
    emitted.clear(); fired.clear()
    monkeypatch.setattr(bench, "train_bench", lambda *a, **k: {"ok": True})
    assert bench.train_bench_guarded(out, None, None, None, 0, 2, None, [], 8000, "x") == {"ok": True}
    assert bench.train_bench_guarded(out, None, None, None, 0, 1, None, [], 8000, "x") == {"ok": True}
    assert emitted == [] and fired == []


def test_merge_strings_helper_is_the_python_loop(native):
    """csrc/swt_pyhost.c swt_py_bpe_merge_strings against the per-merge loop of NaiveBPE.train it stands in for (bpe.py:102-104):
    same strings / index / vocab / merges afterwards, same stop at the first merge whose id is not the expected one"""
    import ctypes as C
    import random

    from subword_tokenizers_amd.tokenizers import _SymbolTable

    host = native.pyhost()
    assert host is not None
    base = native.SYM_BASE
    rng = random.Random(9)

    def python_loop(lefts, rights, first, syms, vocab, merges):
        for i, (left, right) in enumerate(zip(lefts, rights)):
            ls, rs = syms.string(left), syms.string(right)
            joined = ls + rs
            merged = syms.intern(joined)
            vocab.add(joined)
            merges.append((ls, rs))
            if merged != first + i:
                return i, merged
        return -1, None

    for trial in range(200):
        alphabet = [ord(c) for c in "abł中\U0001F600\ud800"][: rng.randrange(1, 7)]
        seed_strings = []
        for _ in range(rng.randrange(0, 6)):
            seed_strings.append("".join(chr(rng.choice(alphabet)) for _ in range(rng.randrange(2, 5))))
        seed_strings = list(dict.fromkeys(seed_strings))
        states = []
        for which in range(2):
            syms = _SymbolTable()
            for s_ in seed_strings:
                syms.intern(s_)
            states.append((syms, set("xyz"), [("x", "y")]))
        n = rng.randrange(0, 40)
        first = base + len(seed_strings)
        lefts, rights = [], []
        for i in range(n):  # ids that exist by the time merge i is reached if no collision happens (else the loop stops anyway)
            pool = alphabet + [base + k for k in range(len(seed_strings) + i)]
            lefts.append(rng.choice(pool)); rights.append(rng.choice(pool))
        want = python_loop(lefts, rights, first, *states[0])
        la, ra = np.array(lefts, dtype=np.uint32), np.array(rights, dtype=np.uint32)
        hit = C.c_uint32()
        syms, vocab, merges = states[1]
        at = host.swt_py_bpe_merge_strings(la.ctypes.data, ra.ctypes.data, n, base, first, syms.strings, syms.index, vocab, merges, C.byref(hit))
        got = (at, hit.value if at >= 0 else None)
        assert got == want, (trial, got, want)
        assert syms.strings == states[0][0].strings and syms.index == states[0][0].index
        assert vocab == states[0][1] and merges == states[0][2]
    with pytest.raises(IndexError):
        la = np.array([base + 99], dtype=np.uint32)
        host.swt_py_bpe_merge_strings(la.ctypes.data, la.ctypes.data, 1, base, base, [], {}, set(), [], C.byref(C.c_uint32()))


def test_decode_ids_large_outputs_equal_the_per_id_spelling(swt, native, ref_dir):
    """FastBPE.decode_ids above 4,096 ids (distinct ids found by a presence map, spelled once, references handed out) against the
    plain per-id form, with and without csrc/swt_pyhost.c; no device needed: the rank table is built on the host"""
    tok = swt.FastBPE()
    tok.load_resources(os.path.join(ref_dir, "resources/pretrained/FastBPE"))
    st = tok._syms
    rng = np.random.default_rng(12)
    pool = np.concatenate([np.arange(97, 123), [0x142, 0x4E2D, 0x1F600], native.SYM_BASE + rng.integers(0, len(st.strings), size=3000)]).astype(np.uint32)
    ids = pool[rng.integers(0, pool.size, size=20000)]
    ids[rng.random(ids.size) < 0.6] |= np.uint32(native.BPE_CONT)
    want = [("##" + st.string(int(t))) if int(t) & native.BPE_CONT else st.string(int(t)) for t in ids]
    assert tok.decode_ids(ids[:100]) == want[:100]
    assert tok.decode_ids(ids) == want
    saved = native._pyhost
    try:
        native._pyhost = False
        assert tok.decode_ids(ids) == want
    finally:
        native._pyhost = saved
    with pytest.raises(IndexError):
        tok.decode_ids(np.full(5000, native.SYM_BASE + len(st.strings), dtype=np.uint32))


def test_sharded_trainer_refuses_a_string_collision_and_a_repeated_pair(native):
    """distributed.ShardedBpeTrainer.train (bpe.py:88-111 over shards): two merges spelling one string (never
    observed on real data) cannot be replayed by a sharded run -- the unsharded class rebuilds from the text, a rank has no way to
    -- so the host loop must say so instead of drifting from the reference; and a pair selected twice means the replicas have
    diverged.  Driven here by a stub engine with the two-method interface (no device)."""
    from subword_tokenizers_amd.distributed import ShardedBpeTrainer

    a, b, c, base = ord("a"), ord("b"), ord("c"), native.SYM_BASE

    class Engine:
        def __init__(self, merges):
            self.merges = merges

        def begin(self):
            return np.array([a, b, c], dtype=np.uint32)

        def run(self, want, first):
            assert first == base
            m = self.merges[:want]
            return (np.array([x for x, _ in m], dtype=np.uint32), np.array([y for _, y in m], dtype=np.uint32),
                    np.array([5] * len(m), dtype=np.uint64))

    # (b,c)->"bc", (a,b)->"ab", (ab,c)->"abc", then (a,bc) spells "abc" again
    tr = ShardedBpeTrainer(Engine([(b, c), (a, b), (base + 1, c), (a, base)]), 0, 2)
    with pytest.raises(RuntimeError, match="already names symbol"):
        tr.train(3 + 4)
    assert tr.merges_list[:3] == [("b", "c"), ("a", "b"), ("ab", "c")]
    tr = ShardedBpeTrainer(Engine([(b, c), (a, b), (b, c)]), 0, 2)
    with pytest.raises(RuntimeError, match="selected twice"):
        tr.train(3 + 3)
    # the ordinary case: the merges come back as strings, the vocabulary grows by one per merge, an early end is noticed
    tr = ShardedBpeTrainer(Engine([(b, c), (a, base)]), 0, 2)
    assert tr.train(3 + 5) == [("b", "c"), ("a", "bc")] and tr.vocab == {"a", "b", "c", "bc", "abc"}



def test_rank_table_placement_and_proper_flag(native, swt, ref_dir):
    """swt_bpe_table_create on the host side (no GPU): the two-choice placement holds every distinct pair exactly once, in a slot
    a device lookup reads (swt_debug_bpe_table_info 4), a later duplicate replaces the earlier entry (bpe.py:257), and the
    'proper' flag -- which lets the word-lane kernel merge one occurrence per round -- is set for trained tables and cleared
    for a table in which a pair ranks BELOW a merge that produces one of its symbols."""
    import ctypes as C

    lib = native.lib()
    lib.swt_debug_bpe_table_info.restype = C.c_int
    lib.swt_debug_bpe_table_info.argtypes = [C.c_void_p, C.c_int]

    def info(tab):
        return [lib.swt_debug_bpe_table_info(tab._h, w) for w in range(5)]

    bpe = swt.FastBPE()
    bpe.load_resources(os.path.join(ref_dir, "resources", "pretrained", "FastBPE"))
    bits, packed, proper, entries, placed = info(bpe._table)
    assert packed == 1 and proper == 1 and placed == 1
    assert entries == len(set(bpe.merges_list)) and (1 << bits) >= 2 * entries
    B = 0x110000
    # (a, b) -> ab at rank 1, but (ab, c) already at rank 0: improper
    t = native.BpeTable([B + 0, ord("a")], [ord("c"), ord("b")], [B + 1, B + 0])
    assert info(t)[2:] == [0, 2, 1]
    t.close()
    # the same pair twice: one entry, the later rank
    t = native.BpeTable([ord("a"), ord("x"), ord("a")], [ord("b"), ord("y"), ord("b")], [B + 0, B + 1, B + 0])
    assert info(t)[2:] == [1, 2, 1]
    t.close()
    # many random pairs over a small alphabet: the placement settles (growing the table when it must) and stays exact
    rng = np.random.default_rng(7)
    n = 50000
    pairs = np.unique(rng.integers(0, 3000, size=(n, 2), dtype=np.uint32), axis=0)
    left = np.where(pairs[:, 0] < 200, pairs[:, 0] + 97, B + pairs[:, 0]).astype(np.uint32)
    right = np.where(pairs[:, 1] < 200, pairs[:, 1] + 97, B + pairs[:, 1]).astype(np.uint32)
    merged = (B + 3000 + np.arange(left.size)).astype(np.uint32)
    t = native.BpeTable(left, right, merged)
    bits, packed, proper, entries, placed = info(t)
    assert entries == left.size and placed == 1 and (1 << bits) >= 2 * entries
    t.close()
    t = native.BpeTable(np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32))
    assert info(t)[3:] == [0, 1]
    t.close()
