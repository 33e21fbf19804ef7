"""the reference's benchmark-suite quality metrics (source/benchmarks.py:7-110, 240-282).
Golden values come from the imported reference (tests/golden/make_golden_metrics.py).  CPU: the formulas on token lists the
oracle produces; `-m gpu`: the batched device path (two encodes + the device token histogram) -- integers exact, floats equal."""
import json
import os

import numpy as np
import pytest


def _corpus(ref_dir, rel):
    with open(os.path.join(ref_dir, rel), encoding="utf-8") as f:
        return json.load(f)


def _check(got, want):
    assert got["counts"] == want["counts"]
    for k in ("avg_tokens_per_sentence", "avg_tokens_per_word", "compression_rate", "normalized_sequence_length",
              "subword_fragmentation_rate", "vocabulary_coverage_rate"):
        assert got[k] == want[k], k  # the same integers through the same expression: the same float
    for k in ("slope", "intercept", "correlation"):
        assert got["zipf"][k] == pytest.approx(want["zipf"][k], rel=1e-12), k


def test_metric_formulas_on_oracle_tokens(swt, oracle, golden, ref_dir):
    """the reference's function surface (same names) on token LISTS, with the oracle as the tokenizer"""
    from subword_tokenizers_amd import metrics as M

    merges = [tuple(p) for p in _corpus(ref_dir, "resources/pretrained/FastBPE/merges.json")]
    vocab = sorted(set(_corpus(ref_dir, "resources/pretrained/FastWordPiece/vocab.json")))
    toks = {"FastBPE": oracle.OracleBPE(merges), "FastWordPiece": oracle.OracleWP(vocab)}
    for want in golden("metrics.json"):
        if want["corpus"].endswith("train-5K.json") and want["model"] == "FastWordPiece":
            continue  # 23 k single-word calls through ctypes: the pan_tadeusz case covers the formulas
        corpus = _corpus(ref_dir, want["corpus"])
        tok = toks[want["model"]]
        inputs = [tok.tokenize(s) for s in corpus]
        words = {w for sent in swt.SubwordTokenizer().preprocessing(corpus) for w, _ in sent}
        by_word = {w: tok.tokenize(w) for w in words}
        chars = sum(len(s.replace(" ", "")) for s in corpus)
        total = sum(len(t) for t in inputs)
        got = {"avg_tokens_per_sentence": M.avg_tokens_per_sentence(inputs), "avg_tokens_per_word": M.avg_tokens_per_word(by_word),
               "compression_rate": M.compression_rate(chars, inputs), "normalized_sequence_length": M.normalized_sequence_length(total, chars),
               "subword_fragmentation_rate": M.subword_fragmentation_rate(by_word), "vocabulary_coverage_rate": M.vocabulary_coverage_rate(by_word),
               "zipf": M.zipf_distribution(inputs),
               "counts": {"sentences": len(corpus), "tokens": total, "chars": chars, "unique_words": len(words),
                          "word_tokens": sum(len(t) for t in by_word.values()), "split_words": sum(len(t) > 1 for t in by_word.values()),
                          "covered_words": sum(len(t) == 1 for t in by_word.values()), "distinct_tokens": len({t for s in inputs for t in s})}}
        _check(got, want)
    assert M.avg_tokens_per_sentence([]) == 0.0 and M.avg_tokens_per_word({}) == 0.0
    assert M.normalized_sequence_length(5, 0) == float("inf") and M.compression_rate(7, []) == float("inf")
    assert M.zipf_from_counts([]) == {"slope": 0.0, "intercept": 0.0, "correlation": 0.0}
    assert M.zipf_from_counts([3]) == {"slope": 0.0, "intercept": float(np.log(3)), "correlation": 0.0}


def test_utf8_from_code_points_round_trip():
    from subword_tokenizers_amd.metrics import _utf8_from_code_points

    words = ["a", "żółć", "中文", "\U0001F600x", "", "é́"]
    cps = np.array([ord(c) for w in words for c in w], dtype=np.uint32)
    off = np.concatenate([[0], np.cumsum([len(w) for w in words])]).astype(np.uint64)
    text, boff = _utf8_from_code_points(cps, off)
    data = text.tobytes()
    assert [data[int(boff[i]):int(boff[i + 1])].decode("utf-8") for i in range(len(words))] == words


@pytest.mark.gpu
def test_quality_metrics_on_the_device(swt, native, golden, ref_dir):
    """quality_metrics(): two batched encodes, the unique words from the device census, the device token histogram"""
    from subword_tokenizers_amd import metrics as M

    if native.device_count() < 1:
        pytest.fail("no HIP device")
    native.init(0)
    bpe = swt.FastBPE()
    bpe.load_resources(os.path.join(ref_dir, "resources/pretrained/FastBPE"))
    wp = swt.FastWP()
    wp.load_resources(os.path.join(ref_dir, "resources/pretrained/FastWordPiece"))
    for want in golden("metrics.json"):
        tok = bpe if want["model"] == "FastBPE" else wp
        _check(M.quality_metrics(tok, _corpus(ref_dir, want["corpus"])), want)
    # the histogram alone, against numpy, on ids with and without the continuation flag and a heavy head
    rng = np.random.default_rng(5)
    ids = (rng.zipf(1.3, size=300000) % 5000).astype(np.uint32)
    ids[rng.random(ids.size) < 0.4] |= np.uint32(native.BPE_CONT)
    counts = native.token_histogram(ids, 5000)
    want = np.bincount((ids & np.uint32(0x7FFFFFFF)).astype(np.int64) + np.where(ids >> np.uint32(31), 5000, 0), minlength=10000)
    assert np.array_equal(counts.astype(np.int64), want)
    assert native.token_histogram(np.zeros(0, dtype=np.uint32), 8).sum() == 0
    with pytest.raises(ValueError):
        native.token_histogram(np.array([9], dtype=np.uint32), 8)
    # the printed report keeps the reference's labels (benchmarks.py:338-346)
    import contextlib
    import io

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        M.benchmarks(bpe, 0, _corpus(ref_dir, "data/pan_tadeusz.json"), pretrained=True, pretrained_path=os.path.join(ref_dir, "resources/pretrained/FastBPE"))
    out = buf.getvalue()
    assert "=== Tokenization Metrics for FastBPE ===" in out and "Average tokens per sentence:        11.24" in out
    assert "Slope:          -0.7836" in out
