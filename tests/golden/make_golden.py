#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (build container only).

The reference (/root/reference) never travels to the GPU box, so everything the parity tests need from
it is produced here and committed as data:

  ref/                          data artefacts copied verbatim from the reference, in the reference's own
                                layout (data/*.json corpora, resources/pretrained/<Model>/ merges+vocab,
                                resources/tests/<Model>/ tutorial KATs, the author-generated pan_tadeusz
                                token lists) -- data, not code
  bpe_train5k_1000.json         FastBPE.train(train-5K, max_vocab=1000): the 922 merges + digests
  bpe_train_micro.json          tie-break / overlap / exhaustion micro-corpora with their merges
  fuzz_bpe.json                 seeded fuzz sentences -> FastBPE.tokenize (pretrained + 922 tables)
  fuzz_wp.json                  seeded fuzz sentences -> FastWP.tokenize (pretrained + tutorial vocab),
                                each run under an alarm; inputs on which the reference does not
                                terminate are recorded as such
  pretok_fuzz.json              seeded strings -> words of SubwordTokenizer.preprocessing
  trie_digest.json              WPTrie_E2E structure: full dump for the tutorial vocab, sha256 for pretrained
  wp_train_micro.json           NaiveWP.train on micro corpora (vocab sets), for the host-side class

Recipe: the reference classes only touch
`self.tokenizer.backend_tokenizer.pre_tokenizer.pre_tokenize_str`, so a two-attribute shim around the
installed wheel's BertPreTokenizer stands in for AutoTokenizer.from_pretrained (which needs the network).

Usage:  python tests/golden/make_golden.py          (takes ~3 minutes; train-5K training dominates)
"""
import hashlib
import json
import os
import random
import shutil
import signal
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)


class _NS:
    pass


def make_shim():
    from tokenizers.pre_tokenizers import BertPreTokenizer

    shim = _NS()
    shim.backend_tokenizer = _NS()
    shim.backend_tokenizer.pre_tokenizer = BertPreTokenizer()
    return shim


class Timeout(Exception):
    pass


def _alarm(signum, frame):
    raise Timeout()


def with_alarm(fn, seconds=0.5):
    signal.signal(signal.SIGALRM, _alarm)
    signal.setitimer(signal.ITIMER_REAL, seconds)
    try:
        return fn()
    finally:
        signal.setitimer(signal.ITIMER_REAL, 0)


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(obj, f, ensure_ascii=False, separators=(",", ":"))
    print("wrote", name, os.path.getsize(path), "bytes")


def sha(obj):
    return hashlib.sha256(json.dumps(obj, ensure_ascii=False).encode("utf-8")).hexdigest()


# --------------------------------------------------------------------------------------------

SPECIALS = (
    list(".,;:!?-()\"'#_$%&*+/<=>@[\\]^`{|}~")
    + list("«»—–„”…§¶¿¡‰†•")
    + list("€£©®°±×÷¢¥")
    + ["́", "̇", "̨"]
    + list(" \t\n\r\x0b\x0c\x1c\x1d\x1e\x1f\x85\xa0       　")
    + ["​", "﻿", "\x00", "İ", "ß", "Σ", "中", "日", "😀", "\U0001D7D8", "ǅ", "ﬁ", "²", "½", "٣"]
    + list("0123456789")
    + list("ĄĆĘŁŃÓŚŹŻABCXYZ")
)
WS_VARIANTS = list(" \t\n\r\x0b\x0c\x1c\x1d\x1e\x1f\x85\xa0       　")


def fuzz_sentences(rng, base, n, alphabet, p_case=0.1):
    out = []
    for k in range(n):
        mode = k % 4
        if mode == 3:  # pure random string over alphabet + letters
            L = rng.randint(0, 60)
            s = "".join(rng.choice(alphabet + list("aąbcćdeęłńóśźżz   ")) for _ in range(L))
        else:
            s = list(rng.choice(base))
            if mode >= 1:
                for _ in range(rng.randint(1, 6)):
                    pos = rng.randint(0, len(s))
                    s.insert(pos, rng.choice(alphabet))
            if mode == 2:
                for _ in range(rng.randint(0, 3)):
                    if s:
                        del s[rng.randint(0, len(s) - 1)]
            s = "".join(c.upper() if rng.random() < p_case else c for c in s)
        out.append(s)
    # hand-picked edge cases
    out += ["", " ", "  \t ", "a", "#", "##", "a ## b", "###", "a##b", "##a", "A_b", "$5+3^2", "a\xa0b", "a\x85b",
            "a b", "a​b", "a﻿b", "a\x00b", "a\x1cb", "İstanbul", "abc€def", "«x»", "x—y", "1,5%",
            "˝zgoda˝", "5×2km", "hello!", "(a", "aaa aaa", "aaaa", "aaaaa", "abababab", "x" * 300,
            "zażółć gęślą jaźń", "ZAŻÓŁĆ GĘŚLĄ JAŹŃ", "ΟΔΥΣΣΕΥΣ ΟΔΟΣ", "straße", "ǅungla", "ﬁn", "áb",
            "słowo " * 40, "...", "a.b.c", "wy-raz", "nie—tak", "pół/na/pół", "e-mail@domena.pl", "100%", "1.000,50",
            " a", "a ", "\na\n", "a\x1fb", "a　b", "!", "!!", "a!", "!a", "#a", "a#", "# #", "## ##", "####"]
    return out


def main():
    t0 = time.time()
    shim = make_shim()
    from source.bpe import FastBPE, NaiveBPE
    from source.utils import SubwordTokenizer
    from source.wordpiece import FastWP, NaiveWP

    print("reference imported in %.1fs" % (time.time() - t0))

    # ---- 1. data artefacts (verbatim copies of JSON data) ----
    ref_dir = os.path.join(HERE, "ref")
    os.makedirs(ref_dir, exist_ok=True)
    copies = {
        "data/train-5K.json": "data/train-5K.json",
        "data/pan_tadeusz.json": "data/pan_tadeusz.json",
        "resources/pretrained/FastBPE/merges.json": "resources/pretrained/FastBPE/merges.json",
        "resources/pretrained/FastWordPiece/vocab.json": "resources/pretrained/FastWordPiece/vocab.json",
        "resources/tests/FastBPE/merges.json": "resources/tests/FastBPE/merges.json",
        "resources/tests/FastWordPiece/vocab.json": "resources/tests/FastWordPiece/vocab.json",
    }
    for src, dst in copies.items():
        os.makedirs(os.path.dirname(os.path.join(ref_dir, dst)), exist_ok=True)
        shutil.copyfile(os.path.join(REF, src), os.path.join(ref_dir, dst))
        os.chmod(os.path.join(ref_dir, dst), 0o644)
    gold = json.load(open(os.path.join(REF, "data/pan_tadeusz.tokens.json"), encoding="utf-8"))
    assert gold["NaiveBPE"] == gold["FastBPE"] and gold["NaiveWordPiece"] == gold["FastWordPiece"]
    # the author-generated token lists (the Naive lists are identical, so one copy of each family is kept)
    dump("ref/data/pan_tadeusz.tokens.json", {"FastBPE": gold["FastBPE"], "FastWordPiece": gold["FastWordPiece"]})

    train5k = json.load(open(os.path.join(REF, "data/train-5K.json"), encoding="utf-8"))
    pan = json.load(open(os.path.join(REF, "data/pan_tadeusz.json"), encoding="utf-8"))

    # ---- 2. BPE training on train-5K (config 1) ----
    bpe = FastBPE(shim)
    t = time.time()
    bpe.train(train5k, 1000)
    t_train = time.time() - t
    merges = [list(p) for p in bpe.merges_list]
    t = time.time()
    toks = [bpe.tokenize(s) for s in train5k]
    t_tok = time.time() - t
    dump("bpe_train5k_1000.json", {
        "max_vocab": 1000, "n_merges": len(merges), "merges": merges,
        "merges_sha256": sha([tuple(m) for m in merges]),
        "tokens_sha256": sha(toks), "n_tokens": sum(map(len, toks)),
        "ref_train_seconds": t_train, "ref_tokenize_seconds": t_tok,
        "tokens_first20": toks[:20],
    })
    bpe_small = bpe

    # ---- 3. micro corpora for training tie-breaks ----
    micro = [
        (["This is a sentence.", "Another example sentence."], 25),
        (["aaa aaa"], 10), (["abc abc ab bc"], 10), (["aaaa aaaa aa"], 12), (["ab ab ba ba"], 8),
        (["abab baba abab"], 10), (["x"], 5), ([""], 5), ([], 5), (["a b c"], 2), (["hello hello world"], 8),
        (["a.b,c a.b"], 9), (["zz zz zzz zzzz"], 6), (["ab cd ab cd ef ef"], 12),
        (["Ala ma kota, a kot ma Alę.", "ALA MA KOTA"], 30), (["aaaaaaaaaaaaaaaa"], 8),
        (["abcabcabc bcabca cabcab"], 14), (["the then they them there the"], 20),
    ]
    out_micro = []
    for corpus, mv in micro:
        m = NaiveBPE(shim)
        m.train(list(corpus), mv)
        out_micro.append({"corpus": corpus, "max_vocab": mv, "merges": [list(p) for p in m.merges_list],
                          "vocab_size": len(m.vocab)})
    # seeded random tiny corpora over small alphabets (trained to exhaustion): tie-break stress
    rng = random.Random(20250629)
    for k in range(120):
        alpha = rng.choice(["ab", "abc", "abcd"])
        nwords = rng.randint(1, 8)
        corpus = [" ".join("".join(rng.choice(alpha) for _ in range(rng.randint(1, 7))) for _ in range(nwords))
                  for _ in range(rng.randint(1, 3))]
        mv = rng.randint(3, 40)
        m = NaiveBPE(shim)
        m.train(list(corpus), mv)
        out_micro.append({"corpus": corpus, "max_vocab": mv, "merges": [list(p) for p in m.merges_list],
                          "vocab_size": len(m.vocab)})
    dump("bpe_train_micro.json", out_micro)

    # ---- 4. fuzz: BPE encode ----
    pre = FastBPE(shim)
    pre.load_resources(os.path.join(REF, "resources/pretrained/FastBPE"))
    rng = random.Random(85000)
    sents = fuzz_sentences(rng, train5k + pan, 600, SPECIALS)
    fb = []
    for s in sents:
        fb.append({"text": s, "pretrained": pre.tokenize(s), "t5k": bpe_small.tokenize(s)})
    words = ["", "a", "ab", "aaa", "aaaa", "nie", "się", "przez", "konstantynopolitańczykowianeczka", "xyzxyzxyz"]
    dump("fuzz_bpe.json", {"sentences": fb,
                           "encode_word": [{"word": w, "pretrained": pre.encode_word(w), "t5k": bpe_small.encode_word(w)}
                                           for w in words]})

    # ---- 5. pre-tokenizer fuzz ----
    base = SubwordTokenizer(shim)
    rng = random.Random(1112064)
    ps = fuzz_sentences(rng, train5k, 300, SPECIALS, p_case=0.3)
    # plus strings sampling the whole code space
    for _ in range(200):
        ps.append("".join(chr(c) for c in (rng.choice([rng.randint(0x20, 0x2FF), rng.randint(0x300, 0xD7FF),
                                                        rng.randint(0xE000, 0xFFFF), rng.randint(0x10000, 0x10FFFF),
                                                        0x20, 0x61]) for _ in range(rng.randint(1, 40)))))
    dump("pretok_fuzz.json", [{"text": s, "words": [w for w, _ in base.preprocessing([s])[0]]} for s in ps])

    # ---- 6. fuzz: FastWP encode (under an alarm) ----
    wp = FastWP(shim)
    wp.load_resources(os.path.join(REF, "resources/pretrained/FastWordPiece"))
    tut = FastWP(shim)
    tut.load_resources(os.path.join(REF, "resources/tests/FastWordPiece"))
    single = sorted(t for t in wp.vocab if len(t) == 1)
    safe_alpha = single + WS_VARIANTS + list("ĄĆĘŁŃÓŚŹŻABCXYZ")
    rng = random.Random(1000000)
    ws = fuzz_sentences(rng, train5k + pan, 400, safe_alpha) + fuzz_sentences(rng, train5k + pan, 200, SPECIALS)

    def run_wp(tok, s):
        try:
            return with_alarm(lambda: tok.tokenize(s))
        except Timeout:
            return "TIMEOUT"
        except IndexError:
            return "INDEXERROR"

    fw = []
    n_to = 0
    for s in ws:
        r1 = run_wp(wp, s)
        r2 = run_wp(tut, s) if len(s) < 80 else None
        n_to += r1 == "TIMEOUT"
        fw.append({"text": s, "pretrained": r1, "tutorial": r2})
    print("wp fuzz: %d inputs, %d timeouts on pretrained" % (len(ws), n_to))
    # vocabularies that exercise the '##' corner (wordpiece.py:260-261) and odd tries
    odd_vocabs = [
        ["a", "b", "##a", "##b", "ab", "##ab", "#", "##"],
        ["a", "b", "##a", "##b", "#", "###"],
        ["a", "b", "##a", "##b", "#", "####"],
        ["a", "b", "##a", "##b"],
        ["a", "##a", "#", "##"],
        ["a", "##a", "#", "###"],
        ["a", "##b", "ab", "abc", "##c", "##bc", "a.b", ".", "##.", "b", "c"],
        ["x", "##x", "x y", " ", "y", "##y"],
        ["un", "##aff", "##able", "##a", "##b", "##l", "##e", "##f", "u", "n", "a", "unaff", "##ffab"],
        [],
    ]
    odd_texts = ["", "a", "ab", "abc", "a b", "## a", "a ## b", "##", "###", "####", "#", "a#", "#a", "a.b", "a.b.c",
                 "a . b", "x y", "x y x", "x  y", "unaffable", "unaffab", "unaffableun", "un aff", "aaa", "abab", "ba",
                 "b a", "a##", "##a", "a ##a", "a ###", "a #### b", "x ", " x", "x y ", "..", ". ."]
    odd = []
    for v in odd_vocabs:
        tok = FastWP(shim)
        tok.vocab = set(v)
        from source.utils import WPTrie_E2E
        tok.vocab_trie = WPTrie_E2E(tok.vocab)
        odd.append({"vocab": sorted(v), "cases": [{"text": s, "tokens": run_wp(tok, s)} for s in odd_texts]})
    dump("fuzz_wp.json", {"sentences": fw, "odd": odd})

    # ---- 7. trie structure ----
    def trie_lines(trie):
        ident = {id(trie.root): "<ROOT>", id(trie.root_p): "<ROOT_P>"}
        lines = []
        stack = [trie.root]
        while stack:
            node = stack.pop()
            for ch, c in node.children.items():
                link = c.failure_link
                link_s = "<NONE>" if link is None else ident.get(id(link), link.chars_seen)
                lines.append([c.chars_seen, int(c.is_end), link_s, list(c.failure_pops)])
                stack.append(c)
        lines.sort(key=lambda r: r[0])
        return lines

    tl = trie_lines(tut.vocab_trie)
    pl = trie_lines(wp.vocab_trie)
    dump("trie_digest.json", {
        "tutorial": {"vocab": sorted(tut.vocab), "nodes": tl},
        "pretrained": {"n_nodes": len(pl), "sha256": sha(pl), "sample": pl[::997]},
    })

    # ---- 8. NaiveWP.train micro (host-side class parity) ----
    wm = []
    for corpus, mv in micro[:12]:
        m = NaiveWP(shim)
        m.train(list(corpus), mv)
        wm.append({"corpus": corpus, "max_vocab": mv, "vocab": sorted(m.vocab)})
    dump("wp_train_micro.json", wm)
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
