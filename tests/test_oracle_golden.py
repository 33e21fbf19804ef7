"""The oracle (oracle/swt_oracle.c) pinned against the reference's own artefacts and against vectors generated
by importing the reference (tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest


def sha(obj):
    return hashlib.sha256(json.dumps(obj, ensure_ascii=False).encode("utf-8")).hexdigest()


def test_class_table_matches_fixture(oracle, golden):
    fx = golden("unicode_classes.json")
    for bit, name in ((1, "bert_ws"), (2, "bert_punct"), (4, "py_space"), (8, "py_alnum")):
        member = np.zeros(0x110000, dtype=bool)
        for lo, hi in fx[name]:
            member[lo:hi + 1] = True
        assert int(member.sum()) == fx["meta"]["counts"][name]
        # spot-check every range edge and a stride through the code space
        probe = sorted({c for lo, hi in fx[name] for c in (lo - 1, lo, hi, hi + 1) if 0 <= c < 0x110000} | set(range(0, 0x110000, 257)))
        for cp in probe:
            assert bool(oracle.lib().orc_class(cp) & bit) == bool(member[cp]), (name, hex(cp))


def test_python_predicates_match_this_interpreter(oracle):
    # py_space / py_alnum were probed from CPython 3.10; the GPU box runs the same interpreter
    for cp in list(range(0, 0x3000)) + list(range(0xF900, 0x10000, 7)) + list(range(0x10000, 0x110000, 997)):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        c = oracle.lib().orc_class(cp)
        assert bool(c & 4) == chr(cp).isspace(), hex(cp)
        assert bool(c & 8) == chr(cp).isalnum(), hex(cp)


def test_pretokenize_fuzz(oracle, golden):
    for case in golden("pretok_fuzz.json"):
        assert oracle.pretokenize(case["text"].lower()) == case["words"], repr(case["text"])


def test_bpe_encode_author_golden(oracle, golden, corpora):
    """reference data/pan_tadeusz.tokens.json: written by the reference's authors (cli.py:266-271)"""
    bpe = oracle.OracleBPE(golden("ref/resources/pretrained/FastBPE/merges.json"))
    assert [bpe.tokenize(s) for s in corpora["pan"]] == corpora["pan_tokens"]["FastBPE"]


def test_wp_encode_author_golden(oracle, golden, corpora):
    wp = oracle.OracleWP(golden("ref/resources/pretrained/FastWordPiece/vocab.json"))
    assert wp.n_nodes == 50173  # 50,172 trie nodes + the detached root_p
    assert [wp.tokenize(s) for s in corpora["pan"]] == corpora["pan_tokens"]["FastWordPiece"]


def test_bpe_encode_fuzz(oracle, golden):
    fz = golden("fuzz_bpe.json")
    pre = oracle.OracleBPE(golden("ref/resources/pretrained/FastBPE/merges.json"))
    t5 = oracle.OracleBPE(golden("bpe_train5k_1000.json")["merges"])
    for c in fz["sentences"]:
        assert pre.tokenize(c["text"]) == c["pretrained"], repr(c["text"])
        assert t5.tokenize(c["text"]) == c["t5k"], repr(c["text"])
    for c in fz["encode_word"]:
        assert pre.encode_word(c["word"]) == c["pretrained"]
        assert t5.encode_word(c["word"]) == c["t5k"]


def test_bpe_train_tutorial_kat(oracle, golden):
    """reference resources/tests/FastBPE/merges.json (README tutorial, max_vocab=25)"""
    tr = oracle.OracleBPETrainer(["This is a sentence.", "Another example sentence."])
    tr.run(25)
    assert [list(p) for p in tr.merges_list] == golden("ref/resources/tests/FastBPE/merges.json")


def test_bpe_train_micro(oracle, golden):
    for c in golden("bpe_train_micro.json"):
        tr = oracle.OracleBPETrainer(c["corpus"])
        tr.run(c["max_vocab"])
        assert [list(p) for p in tr.merges_list] == c["merges"], c["corpus"]
        assert tr.vocab_size == c["vocab_size"]


def test_bpe_train_5k_digests(oracle, golden, corpora):
    """config 1: train-5K, max_vocab=1000 -> 922 merges; digests from SURVEY.md section 8c / BASELINE.md"""
    g = golden("bpe_train5k_1000.json")
    assert g["merges_sha256"] == "f5f4451432124d34d4b1803a7482deebb3ac78d88e891cd6e8f10f5ad8967a44"
    assert g["tokens_sha256"] == "55a20c282ddab78c493886bdea37fd911b418c92c3b3968589fd650da9fb4658"
    tr = oracle.OracleBPETrainer(corpora["t5k"])
    assert (tr.n_words, tr.n_symbols, tr.vocab_size) == (22971, 187885, 78)
    tr.run(1000)
    merges = tr.merges_list
    assert len(merges) == 922 and sha(merges) == g["merges_sha256"]
    bpe = oracle.OracleBPE(merges)
    toks = [bpe.tokenize(s) for s in corpora["t5k"]]
    assert sum(map(len, toks)) == g["n_tokens"] == 168703
    assert sha(toks) == g["tokens_sha256"]


def _run_wp(tok, text):
    try:
        return tok.tokenize(text)
    except RuntimeError:
        return "TIMEOUT"
    except IndexError:
        return "INDEXERROR"


def test_wp_fuzz_including_nontermination(oracle, golden):
    fw = golden("fuzz_wp.json")
    wp = oracle.OracleWP(golden("ref/resources/pretrained/FastWordPiece/vocab.json"))
    tut = oracle.OracleWP(golden("ref/resources/tests/FastWordPiece/vocab.json"))
    n_to = 0
    for c in fw["sentences"]:
        r = _run_wp(wp, c["text"])
        n_to += r == "TIMEOUT"
        assert r == c["pretrained"], repr(c["text"])
        if c["tutorial"] is not None:
            assert _run_wp(tut, c["text"]) == c["tutorial"], repr(c["text"])
    assert n_to > 50  # the fuzz set does exercise the reference's infinite loops


def test_wp_odd_vocabularies(oracle, golden):
    for o in golden("fuzz_wp.json")["odd"]:
        tok = oracle.OracleWP(o["vocab"])
        for c in o["cases"]:
            assert _run_wp(tok, c["text"]) == c["tokens"], (o["vocab"], c["text"])


def test_batch_forms_agree_with_single(oracle, golden, corpora):
    bpe = oracle.OracleBPE(golden("ref/resources/pretrained/FastBPE/merges.json"))
    texts = corpora["pan"][:50] + ["", " ", "a"]
    ids, off = bpe.tokenize_batch_ids(texts)
    for i, t in enumerate(texts):
        assert np.array_equal(ids[int(off[i]):int(off[i + 1])], bpe.tokenize_ids(t))
    wp = oracle.OracleWP(golden("ref/resources/pretrained/FastWordPiece/vocab.json"))
    texts = corpora["pan"][:50] + ["", " ", "hello!", "a ## b"]
    ids, off, st = wp.tokenize_batch_ids(texts)
    for i, t in enumerate(texts):
        one, s1 = wp.tokenize_ids(t)
        assert st[i] == s1
        if s1 == 0:
            assert np.array_equal(ids[int(off[i]):int(off[i + 1])], one)
        else:
            assert off[i] == off[i + 1]


def _wp_order_cases(golden, ref_dir):
    import json
    for c in golden("wp_train_order.json"):
        if "corpus" in c:
            corpus = c["corpus"]
        else:
            with open(os.path.join(os.path.dirname(ref_dir), c["corpus_ref"]["file"]), encoding="utf-8") as f:
                corpus = json.load(f)[: c["corpus_ref"]["first"]]
        yield c, corpus


def test_wp_train_order_matches_reference(oracle, golden, ref_dir):
    """NaiveWP.train (wordpiece.py:29-103): initial symbols, the merge sequence and the final vocabulary of the reference"""
    n = 0
    for c, corpus in _wp_order_cases(golden, ref_dir):
        tr = oracle.OracleWPTrainer(corpus)
        assert tr.vocab_size == len(c["initial"])
        tr.run(c["max_vocab"])
        got = [[l, r] for l, r in tr.merges_list]
        assert got == c["merges"], (c.get("corpus", c.get("corpus_ref")), got[:5], c["merges"][:5])
        vocab = set(c["initial"]) | set(tr.merged_tokens)
        assert sorted(vocab) == c["vocab"]
        assert tr.vocab_size == len(c["vocab"])
        n += len(got)
    assert n > 300


def test_wp_score_is_pythons_int_division(oracle):
    """wordpiece.py:86: freq / (f_l * f_r) on Python ints is the correctly rounded quotient, also past 2^53"""
    import random
    import struct
    rng = random.Random(86)
    cases = [(1, 1, 1), (1, 3, 1), (2, 3, 7), (1, 2**32 - 1, 2**32 - 1), (5, 2**31 + 11, 2**30 + 3), (2**40, 2**50 + 1, 2**13 - 1)]
    for _ in range(4000):
        bits = rng.choice([8, 20, 27, 31, 40, 52, 60, 63])
        fl, fr = rng.randrange(1, 2**bits), rng.randrange(1, 2**rng.choice([8, 20, 27, 31, 40, 52, 60, 63]))
        cnt = rng.randrange(1, min(fl, fr, 2**53) + 1)
        cases.append((cnt, fl, fr))
    for cnt, fl, fr in cases:
        want = struct.unpack("<Q", struct.pack("<d", cnt / (fl * fr)))[0]
        assert oracle.wp_score_bits(cnt, fl, fr) == want, (cnt, fl, fr)


def test_oracle_on_the_headline_corpus_against_the_reference_windows(oracle, golden):
    """S85k-open (bench.py's training corpus): the reference's own FastBPE.train produced these merges when started at merge 0
    (tools/ref_python_baseline.py; the later windows restart from the oracle's state and are checked there, and on the GPU box
    by tests/test_gpu_configs.py where the oracle's full run is affordable)."""
    from subword_tokenizers_amd import synth

    ref = golden("ref_s85k_open_windows.json")
    w0 = ref["windows"][0]
    assert w0["first_merge"] == 0
    orc = oracle.OracleBPETrainer(synth.s85k_open())
    orc.run(ref["max_vocab"], len(w0["merges"]))
    assert [list(m) for m in orc.merges_list] == w0["merges"]

