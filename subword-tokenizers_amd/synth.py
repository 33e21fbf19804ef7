"""Seeded synthetic workloads of SURVEY.md section 8(d) (the real `data/train-85k.json` is absent from the
reference snapshot: /root/reference/.MISSING_LARGE_BLOBS).  Everything derives from committed fixture data
(tests/golden/ref/: train-5K.json, the pretrained merges/vocab) and a seed, so the GPU box regenerates the
same bytes.  Pure Python/numpy; nothing here touches the GPU.

  s85k()            S85k: 85,000 sentences, seed 85000 -- stand-in for train-85k (configs 2)
  v30k()            V30k: pretrained 20k WordPiece vocab + 10,000 new tokens, seed 30000 (config 3)
  wp_corpus()       Zipf(1.1) sentences over the S85k word list restricted to V30k's single-char alphabet
  train_words()     deduplicated word types + Zipf(1.05) frequencies for the training configs (config 4)
"""
import json
import os
import random
from collections import Counter
from functools import lru_cache

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(os.path.dirname(_HERE), "tests", "golden", "ref")


def _load(rel):
    with open(os.path.join(REF, rel), encoding="utf-8") as f:
        return json.load(f)


@lru_cache(maxsize=None)
def train5k():
    return _load("data/train-5K.json")


@lru_cache(maxsize=None)
def pretrained_merges():
    return [tuple(p) for p in _load("resources/pretrained/FastBPE/merges.json")]


@lru_cache(maxsize=None)
def pretrained_vocab():
    return _load("resources/pretrained/FastWordPiece/vocab.json")


def _is_word_char(ch):
    return ch.isalnum()


@lru_cache(maxsize=None)
def _t5k_stats():
    """Lowercase words (maximal alnum runs) with counts, punctuation marks with counts, sentence lengths."""
    words = Counter()
    puncts = Counter()
    lengths = []
    for s in train5k():
        s = s.lower()
        lengths.append(len(s))
        cur = []
        for ch in s:
            if _is_word_char(ch):
                cur.append(ch)
            else:
                if cur:
                    words["".join(cur)] += 1
                    cur = []
                if not ch.isspace():
                    puncts[ch] += 1
        if cur:
            words["".join(cur)] += 1
    return words, puncts, lengths


def _trigram_model(words):
    """char trigram counts with ^^ / $ padding, weighted by word frequency"""
    model = {}
    for w, f in words.items():
        p = "^^" + w + "$"
        for i in range(len(p) - 2):
            model.setdefault(p[i:i + 2], Counter())[p[i + 2]] += f
    return {ctx: (list(c.keys()), np.cumsum(list(c.values())).tolist()) for ctx, c in model.items()}


def _sample_word(rng, model, max_len=24):
    ctx, out = "^^", []
    while len(out) < max_len:
        chars, cum = model[ctx]
        ch = rng.choices(chars, cum_weights=cum)[0]
        if ch == "$":
            break
        out.append(ch)
        ctx = ctx[1] + ch
    return "".join(out)


@lru_cache(maxsize=None)
def s85k_lexicon(n_novel=60000, seed=85000):
    """(words, cumulative weights): train-5K's word types by frequency + trigram-sampled novel types."""
    words, _, _ = _t5k_stats()
    rng = random.Random(seed)
    model = _trigram_model(words)
    known = set(words)
    novel = []
    seen = set()
    while len(novel) < n_novel:
        w = _sample_word(rng, model)
        if w and w not in known and w not in seen:
            seen.add(w)
            novel.append(w)
    total = sum(words.values())
    lex = list(words.keys()) + novel
    # 85 % of the mass on seen types (by their frequency), 15 % spread evenly over the novel types
    weights = [0.85 * f / total for f in words.values()] + [0.15 / n_novel] * n_novel
    return lex, np.cumsum(weights).tolist()


def sentences(n, seed, lexicon=None):
    """n sentences shaped like train-5K (length distribution, punctuation set), words from the lexicon."""
    lex, cum = lexicon or s85k_lexicon()
    _, puncts, lengths = _t5k_stats()
    rng = random.Random(seed)
    pun_chars = list(puncts.keys())
    pun_cum = np.cumsum(list(puncts.values())).tolist()
    words_total = sum(_t5k_stats()[0].values())
    p_punct = sum(puncts.values()) / max(words_total, 1)
    targets = rng.choices(lengths, k=n)
    need = int(sum(targets) / 5.5) + 16 * n + 1024
    pool = rng.choices(lex, cum_weights=cum, k=need)
    ppool = rng.choices(pun_chars, cum_weights=pun_cum, k=need)
    coin = [rng.random() for _ in range(need)]
    out = []
    k = 0
    for t in targets:
        parts = []
        ln = 0
        while ln < t:
            w = pool[k]
            if coin[k] < p_punct:
                w += ppool[k]
            k += 1
            parts.append(w)
            ln += len(w) + 1
        s = " ".join(parts)
        out.append(s[:1].upper() + s[1:])
    return out


@lru_cache(maxsize=2)
def s85k(n=85000, seed=85000):
    """S85k: the declared stand-in for data/train-85k.json."""
    return sentences(n, seed)


# ---- S85k-open: SURVEY.md section 8(d)2 to the letter -------------------------------------------------------------------
# EVERY word is drawn from the char-trigram model fit on train-5K's lowercase words (frequency-weighted), so unseen word
# types keep appearing as the corpus grows (an open vocabulary); sentence lengths and punctuation marks are train-5K's
# empirical distributions.  Vectorised: all words of the corpus advance one character per numpy step.

@lru_cache(maxsize=None)
def _trigram_tables():
    """Dense form of the trigram model: alphabet (index 0 = '^' padding, last = '$' end), cum[ctx0 * A + ctx1, next]."""
    words, _, _ = _t5k_stats()
    chars = sorted({ch for w in words for ch in w})
    alpha = ["^"] + chars + ["$"]
    idx = {c: i for i, c in enumerate(alpha)}
    A = len(alpha)
    cnt = np.zeros((A * A, A), dtype=np.float64)
    for w, f in words.items():
        p = [0, 0] + [idx[c] for c in w] + [A - 1]
        for i in range(len(p) - 2):
            cnt[p[i] * A + p[i + 1], p[i + 2]] += f
    tot = cnt.sum(axis=1, keepdims=True)
    tot[tot == 0] = 1.0
    cum = np.cumsum(cnt / tot, axis=1)
    cum[:, -1] = 1.0
    return alpha, cum


def _sample_words_trigram(rng, n, max_len=24):
    """n words (list[str], none empty) from the trigram model."""
    alpha, cum = _trigram_tables()
    A = len(alpha)
    out = np.zeros((n, max_len), dtype=np.int32)
    lens = np.zeros(n, dtype=np.int32)
    c0 = np.zeros(n, dtype=np.int64)
    c1 = np.zeros(n, dtype=np.int64)
    alive = np.arange(n)
    flat = (cum + np.arange(A * A, dtype=np.float64)[:, None]).ravel()
    for step in range(max_len):
        if alive.size == 0:
            break
        # inverse CDF of every live word's context in one searchsorted: row r of cum shifted to [r, r + 1]
        ctx = c0[alive] * A + c1[alive]
        nxt = np.searchsorted(flat, ctx + rng.random(alive.size) * (1.0 - 1e-12), side="right") - ctx * A
        if step == 0:  # an empty word is not a word: redraw the end marker as the most likely first character
            first = cum[0]
            best = int(np.argmax(np.diff(np.concatenate([[0.0], first[:-1]]))))
            nxt[nxt >= A - 1] = best
        nxt = np.minimum(nxt, A - 1)
        go = nxt < A - 1
        sel = alive[go]
        out[sel, step] = nxt[go]
        lens[sel] = step + 1
        c0[sel] = c1[sel]
        c1[sel] = nxt[go]
        alive = sel
    table = np.array([ord(c) for c in alpha], dtype=np.uint32)
    cps = table[out]
    blob = cps.astype("<u4").tobytes().decode("utf-32-le")
    return [blob[i * max_len:i * max_len + int(lens[i])] for i in range(n)]


def sentences_open(n, seed):
    """n sentences: train-5K's sentence-length distribution and punctuation marks, every word trigram-sampled."""
    _, puncts, lengths = _t5k_stats()
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    pun_chars = list(puncts.keys())
    pun_cum = np.cumsum(list(puncts.values())).tolist()
    words_total = sum(_t5k_stats()[0].values())
    p_punct = sum(puncts.values()) / max(words_total, 1)
    targets = rng.choices(lengths, k=n)
    need = int(sum(targets) / 4.8) + 2 * n + 1024
    pool = _sample_words_trigram(nrng, need)
    ppool = rng.choices(pun_chars, cum_weights=pun_cum, k=need)
    coin = nrng.random(need)
    out = []
    k = 0
    for t in targets:
        parts = []
        ln = 0
        while ln < t:
            if k == len(pool):  # the estimate fell short: top the three pools up
                more = max(need // 8, 1024)
                pool += _sample_words_trigram(nrng, more)
                ppool += rng.choices(pun_chars, cum_weights=pun_cum, k=more)
                coin = np.concatenate([coin, nrng.random(more)])
            w = pool[k]
            if coin[k] < p_punct:
                w += ppool[k]
            k += 1
            parts.append(w)
            ln += len(w) + 1
        s = " ".join(parts)
        out.append(s[:1].upper() + s[1:])
    return out


@lru_cache(maxsize=2)
def s85k_open(n=85000, seed=85000):
    """S85k-open: the section 8(d)2 stand-in for data/train-85k.json with an OPEN vocabulary (every word trigram-sampled)."""
    return sentences_open(n, seed)


def word_stats(sents):
    """(words, distinct words) of the pre-tokenizer split approximated on the host: maximal alnum runs + single marks."""
    c = Counter()
    for s in sents:
        cur = []
        for ch in s.lower():
            if ch.isalnum():
                cur.append(ch)
            else:
                if cur:
                    c["".join(cur)] += 1
                    cur = []
                if not ch.isspace():
                    c[ch] += 1
        if cur:
            c["".join(cur)] += 1
    return sum(c.values()), len(c)


@lru_cache(maxsize=None)
def v30k(seed=30000, n_new=10000):
    """V30k: the pretrained 20,000-token vocabulary + n_new prefixes / '##' suffixes of S85k words (config 3)."""
    base = pretrained_vocab()
    have = set(base)
    rng = random.Random(seed)
    lex, _ = s85k_lexicon()
    new = []
    while len(new) < n_new:
        w = rng.choice(lex)
        if len(w) < 3:
            continue
        cut = rng.randint(2, len(w) - 1)
        tok = w[:cut] if rng.random() < 0.5 else "##" + w[cut:]
        if tok not in have:
            have.add(tok)
            new.append(tok)
    return base + new


def wp_corpus(n_sent, seed=1000000, vocab=None, zipf_a=1.1, max_bytes=512):
    """Packed UTF-8 (uint8) + offsets (uint64) of n_sent sentences: Zipf(zipf_a) over the S85k word list restricted
    to characters that are single-char tokens of `vocab` (so the reference's FastWP terminates on every sentence)."""
    vocab = vocab or v30k()
    single = {t for t in vocab if len(t) == 1}
    lex, _ = s85k_lexicon()
    words = [w for w in lex if all(ch in single for ch in w)]
    wb = [w.encode("utf-8") for w in words]
    wlen = np.fromiter(map(len, wb), dtype=np.int64, count=len(wb))
    woff = np.zeros(len(wb) + 1, dtype=np.int64)
    np.cumsum(wlen, out=woff[1:])
    blob = np.frombuffer(b"".join(wb), dtype=np.uint8)
    rng = np.random.default_rng(seed)
    per = rng.integers(4, 28, size=n_sent)  # words per sentence
    n_words = int(per.sum())
    ranks = rng.zipf(zipf_a, size=n_words)
    idx = (ranks - 1) % len(wb)
    lens = wlen[idx]
    step = lens + 1  # word + one space
    dst = np.cumsum(step) - step
    total = int(step.sum())
    out = np.full(total, 0x20, dtype=np.uint8)
    intra = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(np.cumsum(lens) - lens, lens)
    out[np.repeat(dst, lens) + intra] = blob[np.repeat(woff[idx], lens) + intra]
    first = np.cumsum(per) - per
    off = np.zeros(n_sent + 1, dtype=np.uint64)
    off[:-1] = dst[first]
    off[-1] = total
    assert int((off[1:] - off[:-1]).max()) <= max_bytes
    return out, off


def unpack(text_u8, off, lo=0, hi=None):
    """sentences lo..hi of a packed corpus as Python strings (for oracle subsamples)"""
    hi = len(off) - 1 if hi is None else hi
    b = text_u8.tobytes()
    return [b[int(off[i]):int(off[i + 1])].decode("utf-8") for i in range(lo, hi)]


def train_words(n_types, seed, zipf_a=1.05, total_tokens=None, min_len=2, max_len=16):
    """Deduplicated training words in the reference's formulation (bpe.py:73-81): unique word types in
    first-occurrence order with Zipf frequencies.  Types come from a char-bigram model over the pretrained
    124-character alphabet.  Returns (symbols uint32, word_off uint64[n+1], freq uint32)."""
    alphabet = sorted({ch for t in pretrained_vocab() for ch in t if ch.isalnum()})
    cps = np.array([ord(c) for c in alphabet], dtype=np.uint32)
    rng = np.random.default_rng(seed)
    A = len(alphabet)
    # bigram transition: a random sparse-ish row-stochastic matrix (seeded), so pairs have skewed counts
    trans = rng.dirichlet(np.full(A, 0.08), size=A)
    cum = np.cumsum(trans, axis=1)
    lens = rng.integers(min_len, max_len + 1, size=n_types)
    total = int(lens.sum())
    off = np.zeros(n_types + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    state = rng.integers(0, A, size=n_types)
    sym = np.empty(total, dtype=np.uint32)
    starts = off[:-1].astype(np.int64)
    alive = np.arange(n_types)
    for step in range(max_len):
        sel = alive[lens[alive] > step]
        if sel.size == 0:
            break
        sym[starts[sel] + step] = cps[state[sel]]
        u = rng.random(sel.size)
        nxt = (cum[state[sel]] < u[:, None]).sum(axis=1)
        state[sel] = np.minimum(nxt, A - 1)
        alive = sel
    ranks = np.arange(1, n_types + 1, dtype=np.float64)
    w = ranks ** (-zipf_a)
    total_tokens = total_tokens or 50 * n_types
    freq = np.maximum(1, np.floor(w / w.sum() * total_tokens)).astype(np.uint32)
    return sym, off, freq
