"""The reference's benchmark-suite quality metrics, computed from token IDS that stay on the device until they are counts
(citations relative to /root/reference).

`source/benchmarks.py` tokenizes the test corpus into Python lists of strings, one `tokenize` call per sentence and one per
unique word, and then counts them (benchmarks.py:7-110, 240-282, 331-337).  Here the same quantities come from two batched
encodes (`encode_ids_batch` of the sentences, and of the unique words), the CSR offsets, and a device-side histogram of the
token ids (`swt_token_histogram`):

    avg_tokens_per_sentence        tokens / sentences                       benchmarks.py:7-21
    avg_tokens_per_word            word tokens / unique words               benchmarks.py:24-38
    normalized_sequence_length     tokens / non-space characters            benchmarks.py:41-52
    subword_fragmentation_rate     unique words in > 1 token, percent       benchmarks.py:55-72
    vocabulary_coverage_rate       unique words in exactly 1 token, percent benchmarks.py:75-92
    compression_rate               non-space characters / tokens            benchmarks.py:95-110
    zipf_distribution              log-log regression of count on rank      benchmarks.py:240-282

The integers are exact; every float is computed from them with the reference's own expression in the reference's order of
operations (Python `int / int`, sequential `sum`), so the results are the reference's floats bit for bit.  The function names
and signatures of the reference are kept for callers that already hold token lists (they accept lists of strings OR the id
arrays this package produces); `quality_metrics` is the batched path.
"""
import math
import os
from typing import Any, Dict, List

import numpy as np

from . import _native as N

__all__ = ["avg_tokens_per_sentence", "avg_tokens_per_word", "normalized_sequence_length", "subword_fragmentation_rate",
           "vocabulary_coverage_rate", "compression_rate", "zipf_distribution", "zipf_from_counts", "quality_metrics",
           "token_sequence_equivalence", "tokenization_performance", "training_performance", "benchmarks"]


# ---- the reference's function surface (benchmarks.py:7-110) ---------------------------------------------------------------

def avg_tokens_per_sentence(tokenized_inputs) -> float:
    if not len(tokenized_inputs):
        return 0.0
    return sum(len(t) for t in tokenized_inputs) / len(tokenized_inputs)


def avg_tokens_per_word(tokenized_words: Dict[str, Any]) -> float:
    if not tokenized_words:
        return 0.0
    return sum(len(t) for t in tokenized_words.values()) / len(tokenized_words)


def normalized_sequence_length(total_tokens: int, total_chars: int) -> float:
    return total_tokens / total_chars if total_chars else float("inf")


def subword_fragmentation_rate(tokenized_words: Dict[str, Any]) -> float:
    if not tokenized_words:
        return 0.0
    return sum(1 for t in tokenized_words.values() if len(t) > 1) / len(tokenized_words) * 100


def vocabulary_coverage_rate(tokenized_words: Dict[str, Any]) -> float:
    if not tokenized_words:
        return 0.0
    return sum(1 for t in tokenized_words.values() if len(t) == 1) / len(tokenized_words) * 100


def compression_rate(total_chars: int, tokenized_inputs) -> float:
    total = sum(len(t) for t in tokenized_inputs)
    return total_chars / total if total else float("inf")


def zipf_from_counts(counts) -> Dict[str, float]:
    """benchmarks.py:255-282 on the token frequencies themselves (any order; zeros are ignored)."""
    freqs = sorted((int(c) for c in counts if c), reverse=True)  # most_common(): count descending
    ranks = range(1, len(freqs) + 1)
    log_ranks = [math.log(r) for r in ranks]
    log_freqs = [math.log(f) for f in freqs]
    n = len(freqs)
    if not n:
        return {"slope": 0.0, "intercept": 0.0, "correlation": 0.0}
    mean_r = sum(log_ranks) / n
    mean_f = sum(log_freqs) / n
    cov = sum((x - mean_r) * (y - mean_f) for x, y in zip(log_ranks, log_freqs))
    var_r = sum((x - mean_r) ** 2 for x in log_ranks)
    var_f = sum((y - mean_f) ** 2 for y in log_freqs)
    slope = cov / var_r if var_r else 0.0
    intercept = mean_f - slope * mean_r
    corr = cov / math.sqrt(var_r * var_f) if var_r and var_f else 0.0
    return {"slope": slope, "intercept": intercept, "correlation": corr}


def zipf_distribution(tokenized_inputs) -> Dict[str, float]:
    """benchmarks.py:240-282 for lists of token strings (the reference's input)."""
    from collections import Counter

    return zipf_from_counts(Counter(t for s in tokenized_inputs for t in s).values())


# ---- the batched path: ids, offsets, device histogram --------------------------------------------------------------------

def _utf8_from_code_points(cps: np.ndarray, off: np.ndarray):
    """code points (uint32) with CSR offsets -> (UTF-8 bytes uint8, byte offsets uint64): numpy, no per-word Python"""
    cp = cps.astype(np.int64)
    nb = np.where(cp < 0x80, 1, np.where(cp < 0x800, 2, np.where(cp < 0x10000, 3, 4)))
    boff = np.zeros(cp.size + 1, dtype=np.int64)
    np.cumsum(nb, out=boff[1:])
    out = np.zeros(int(boff[-1]), dtype=np.uint8)
    p = boff[:-1]
    m = nb == 1
    out[p[m]] = cp[m]
    m = nb == 2
    out[p[m]] = 0xC0 | (cp[m] >> 6)
    out[p[m] + 1] = 0x80 | (cp[m] & 0x3F)
    m = nb == 3
    out[p[m]] = 0xE0 | (cp[m] >> 12)
    out[p[m] + 1] = 0x80 | ((cp[m] >> 6) & 0x3F)
    out[p[m] + 2] = 0x80 | (cp[m] & 0x3F)
    m = nb == 4
    out[p[m]] = 0xF0 | (cp[m] >> 18)
    out[p[m] + 1] = 0x80 | ((cp[m] >> 12) & 0x3F)
    out[p[m] + 2] = 0x80 | ((cp[m] >> 6) & 0x3F)
    out[p[m] + 3] = 0x80 | (cp[m] & 0x3F)
    return out, boff[off.astype(np.int64)].astype(np.uint64)


def unique_words_packed(corpus: List[str]):
    """{word for sentence in preprocessing(corpus) for word, _ in sentence} (benchmarks.py:335) without leaving the device's
    word census: -> (UTF-8 bytes, byte offsets) of the distinct words, in first-occurrence order."""
    text, off = N.pack_and_lower(corpus)
    tr = N.BpeTrainer.from_text(text, off)  # utils.py:27 split + Counter of the words, on the device
    try:
        cps, woff, _freq = tr.export()
    finally:
        tr.close()
    wtext, wboff = _utf8_from_code_points(cps, woff)
    lens = np.diff(wboff.astype(np.int64))
    if lens.size and int(lens.max()) >= 255:
        # the census never matches words of 255 bytes or more against each other (swt_words.hip): finish the set here
        data = wtext.tobytes()
        seen, keep = set(), np.ones(lens.size, dtype=bool)
        for i in np.flatnonzero(lens >= 255):
            w = data[int(wboff[i]):int(wboff[i + 1])]
            if w in seen:
                keep[i] = False
            seen.add(w)
        if not keep.all():
            parts = [data[int(wboff[i]):int(wboff[i + 1])] for i in np.flatnonzero(keep)]
            wtext = np.frombuffer(b"".join(parts), dtype=np.uint8)
            wboff = np.zeros(len(parts) + 1, dtype=np.uint64)
            np.cumsum(np.fromiter(map(len, parts), dtype=np.uint64, count=len(parts)), out=wboff[1:])
    return wtext, wboff


def _encode_packed(tokenizer, text_u8, off):
    """ids + offsets of a packed (already lowercased) batch through the tokenizer's device handle"""
    if hasattr(tokenizer, "_trie") and tokenizer._trie is not None:  # FastWP
        ids, ooff, status = tokenizer._trie.encode(text_u8, off)
        bad = np.flatnonzero(status)
        if bad.size:
            i = int(bad[0])
            tokenizer._raise_for_status(int(status[i]), text_u8[int(off[i]):int(off[i + 1])].tobytes().decode("utf-8", "replace"))
        n_vocab = len(tokenizer._tokens)
        if ids.size and int(ids.max()) > n_vocab + 1:
            raise NotImplementedError("a multi-token NaiveWP.encode_word('##') corner (wordpiece.py:260-261) is not counted on the device")
        return ids, ooff, n_vocab + 3
    table = tokenizer._ensure_table()  # FastBPE
    ids, ooff = table.encode(text_u8, off)
    return ids, ooff, N.SYM_BASE + len(tokenizer._syms.strings) + 1


def quality_metrics(tokenizer, test_corpus: List[str]) -> Dict[str, Any]:
    """Every tokenization metric of benchmarks() (benchmarks.py:331-346 and 354-357) for a FastBPE / FastWP of this package:
    two batched encodes and one device histogram instead of a Python call per sentence and per unique word."""
    if not isinstance(test_corpus, list) or not all(isinstance(s, str) for s in test_corpus):
        raise TypeError("Text must be a string.")
    n_sent = len(test_corpus)
    text, off = N.pack_and_lower(test_corpus)
    ids, _ooff, id_cap = _encode_packed(tokenizer, text, off)
    total_tokens = int(ids.size)
    wtext, wboff = unique_words_packed(test_corpus)
    _wids, woff_tok, _ = _encode_packed(tokenizer, wtext, wboff)
    per_word = np.diff(woff_tok.astype(np.int64))
    n_words = int(per_word.size)
    word_tokens, split, covered = int(per_word.sum()), int((per_word > 1).sum()), int((per_word == 1).sum())
    total_chars = sum(len(s) - s.count(" ") for s in test_corpus)  # len(s.replace(' ', '')), benchmarks.py:336
    counts = N.token_histogram(ids, id_cap)
    return {
        "avg_tokens_per_sentence": total_tokens / n_sent if n_sent else 0.0,
        "avg_tokens_per_word": word_tokens / n_words if n_words else 0.0,
        "compression_rate": total_chars / total_tokens if total_tokens else float("inf"),
        "normalized_sequence_length": total_tokens / total_chars if total_chars else float("inf"),
        "subword_fragmentation_rate": split / n_words * 100 if n_words else 0.0,
        "vocabulary_coverage_rate": covered / n_words * 100 if n_words else 0.0,
        "zipf": zipf_from_counts(counts[counts > 0]),
        "counts": {"sentences": n_sent, "tokens": total_tokens, "chars": total_chars, "unique_words": n_words, "word_tokens": word_tokens,
                   "split_words": split, "covered_words": covered, "distinct_tokens": int((counts > 0).sum())},
    }


# ---- timing (benchmarks.py:186-237) and the printed report (benchmarks.py:285-434) -----------------------------------------

def tokenization_performance(tokenizer: Any, input: List[str]) -> Dict[str, float]:
    """benchmarks.py:186-219 with the device's batch call in place of the per-sentence loop"""
    from timeit import default_timer as timer

    start = timer()
    if hasattr(tokenizer, "encode_ids_batch"):
        n_tokens = int(tokenizer.encode_ids_batch(input)[0].size)
    else:
        n_tokens = sum(len(tokenizer.tokenize(s)) for s in input)
    total_time = timer() - start
    return {"total_time_s": total_time, "throughput_tokens_per_s": n_tokens / total_time if total_time > 0 else float("inf"),
            "avg_latency_s": total_time / len(input) if input else 0.0}


def training_performance(tokenizer: Any, test_corpus: List[str], max_vocab_size: int) -> Dict[str, float]:
    """benchmarks.py:222-237"""
    from timeit import default_timer as timer

    start = timer()
    tokenizer.train(test_corpus, max_vocab_size)
    return {"train_time_s": timer() - start}


def _strip(tokens):
    return [t[2:] if t.startswith("##") else t for t in tokens]


def token_sequence_equivalence(tokenizer1: Any, tokenizer2: Any, input: List[str]):
    """benchmarks.py:113-183: positional / unordered / per-word agreement of two tokenizers.  The token lists come from the
    batch calls where a class has them (one call for the sentences, one for all the words); the counting is the reference's."""
    from collections import Counter

    def many(tok, texts):
        return tok.tokenize_batch(list(texts)) if hasattr(tok, "tokenize_batch") and len(texts) > 1 else [tok.tokenize(t) for t in texts]

    sents1, sents2 = many(tokenizer1, input), many(tokenizer2, input)
    words = [w for s in input for w in s.split()]
    uniq = sorted(set(words))
    w1 = dict(zip(uniq, many(tokenizer1, uniq)))
    w2 = dict(zip(uniq, many(tokenizer2, uniq)))
    total_pos_matches = total_positions = total_unordered_matches = total_word_matches = 0
    for raw1, raw2 in zip(sents1, sents2):
        t1, t2 = _strip(raw1), _strip(raw2)
        n = min(len(t1), len(t2))
        total_pos_matches += sum(1 for i in range(n) if t1[i] == t2[i])
        total_positions += n
        f1, f2 = Counter(t1), Counter(t2)
        total_unordered_matches += sum(min(f1[t], f2[t]) for t in (f1.keys() & f2.keys()))
    for w in words:
        if set(_strip(w1[w])) & set(_strip(w2[w])):
            total_word_matches += 1
    total_words = len(words)
    return (total_pos_matches, total_positions, (total_pos_matches / total_positions * 100) if total_positions else 0.0,
            total_unordered_matches, (total_unordered_matches / total_positions * 100) if total_positions else 0.0,
            total_word_matches, total_words, (total_word_matches / total_words * 100) if total_words else 0.0)


def _report_pretrained(tokenizer, name, test_corpus):
    if hasattr(tokenizer, "encode_ids_batch"):
        m = quality_metrics(tokenizer, test_corpus)
    else:  # the Naive classes: the reference's own per-call path (benchmarks.py:333-337)
        inputs = [tokenizer.tokenize(s) for s in test_corpus]
        words = {w for sent in tokenizer.preprocessing(test_corpus) for w, _ in sent}
        by_word = {w: tokenizer.tokenize(w) for w in words}
        chars = sum(len(s.replace(" ", "")) for s in test_corpus)
        total = sum(len(t) for t in inputs)
        m = {"avg_tokens_per_sentence": avg_tokens_per_sentence(inputs), "avg_tokens_per_word": avg_tokens_per_word(by_word),
             "compression_rate": compression_rate(chars, inputs), "normalized_sequence_length": normalized_sequence_length(total, chars),
             "subword_fragmentation_rate": subword_fragmentation_rate(by_word), "vocabulary_coverage_rate": vocabulary_coverage_rate(by_word),
             "zipf": zipf_distribution(inputs)}
    print(f"=== Tokenization Metrics for {name} ===")
    print(f"Average tokens per sentence:        {m['avg_tokens_per_sentence']:.2f}")
    print(f"Average tokens per word:            {m['avg_tokens_per_word']:.2f}")
    print(f"Compression rate (chars per token): {m['compression_rate']:.2f}")
    print(f"Normalized sequence length:         {m['normalized_sequence_length']:.4f}")
    print(f"Subword fragmentation rate:         {m['subword_fragmentation_rate']:.2f}%")
    print(f"Vocabulary coverage rate:           {m['vocabulary_coverage_rate']:.2f}%")
    print("\n=== Tokenization Performance ===")
    perf = tokenization_performance(tokenizer, test_corpus)
    print(f"Total time:     {perf['total_time_s']:.4f}s")
    print(f"Throughput:     {perf['throughput_tokens_per_s']:.2f} tokens/s")
    print(f"Avg. latency:   {perf['avg_latency_s']:.6f}s per sentence")
    print("\n=== Zipf Distribution Fit ===")
    print(f"Slope:          {m['zipf']['slope']:.4f}")
    print(f"Intercept:      {m['zipf']['intercept']:.4f}")
    print(f"Correlation:    {m['zipf']['correlation']:.4f}")


def benchmarks(tokenizer: Any, max_vocab_size: int, test_corpus: List[str], train_corpus: List[str] = [], pretrained: bool = False,
               pretrained_path: str = "", reference_tokenizers: List[Any] = [], compare_only: bool = False,
               reference_names: List[str] = None, resources_root: str = "") -> None:
    """The printed report of benchmarks.py:285-434 (same labels and formats): tokenization metrics, tokenization performance
    and the Zipf fit when `pretrained` (token-sequence equivalence only with `compare_only`), training time otherwise.
    `pretrained_path` is the primary tokenizer's resource directory; the others load from resources_root/<their name>."""
    name = tokenizer.__class__.__name__
    names = list(reference_names or [t.__class__.__name__ for t in reference_tokenizers])
    if pretrained and compare_only:
        if not reference_tokenizers:
            print("No reference tokenizers provided for comparison.")
            return
        for other, name2 in zip(reference_tokenizers, names):
            (pos, positions, pos_rate, unordered, unordered_rate, word_matches, total_words, word_rate) = token_sequence_equivalence(
                tokenizer, other, test_corpus)
            print(f"=== Token Sequence Equivalence ({name} vs {other.__class__.__name__}) ===")
            print(f"Positional match rate: {pos_rate:.2f}% ({pos}/{positions})")
            print(f"Unordered match rate:  {unordered_rate:.2f}% ({unordered}/{positions})")
            print(f"Word match rate:       {word_rate:.2f}% ({word_matches}/{total_words})")
        return
    if pretrained:
        if pretrained_path:
            tokenizer.load_resources(pretrained_path)
        _report_pretrained(tokenizer, name, test_corpus)
        for other, name2 in zip(reference_tokenizers, names):
            if resources_root:
                other.load_resources(os.path.join(resources_root, name2))
            print()
            _report_pretrained(other, other.__class__.__name__, test_corpus)
    else:
        if not train_corpus:
            raise ValueError("train_corpus is required for training metrics.")
        perf = training_performance(tokenizer, train_corpus, max_vocab_size)
        print(f"=== Training Performance for {name} ===")
        print(f"Training time:  {perf['train_time_s']:.4f}s")
        for other in reference_tokenizers:
            perf2 = training_performance(other, train_corpus, max_vocab_size)
            print(f"\n=== Training Performance for {other.__class__.__name__} ===")
            print(f"Training time:  {perf2['train_time_s']:.4f}s")
