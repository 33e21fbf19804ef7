#!/usr/bin/env python3
"""Golden vectors for NaiveWP.train (wordpiece.py:29-103), made by IMPORTING THE REFERENCE (build container only).

The reference keeps only the final vocabulary SET; the parity tests want the sequence of merges, so this harness
wraps two things around the unmodified class: `_replace_pair` is observed (its first call of an iteration carries the
pair that iteration chose) and the module's progress bar is replaced by a counter that marks the iteration boundary.

  wp_train_order.json   [{corpus | corpus_ref, max_vocab, initial (sorted initial symbols), merges [[left, right], ...],
                          vocab (sorted final vocabulary)}]

Same shim recipe as make_golden.py.  Usage: python tests/golden/make_golden_wptrain.py  (~2 min)
"""
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
from make_golden import make_shim, dump  # noqa: E402


def main():
    t0 = time.time()
    import source.wordpiece as W

    state = {"merges": None, "fresh": True}

    class Bar:
        def __init__(self, *a, **k):
            pass

        def update(self, n=1):
            state["fresh"] = True  # the next _replace_pair call belongs to a new iteration... of the NEXT pass

        def close(self):
            pass

    W.tqdm = Bar
    orig = W.NaiveWP._replace_pair

    def spy(self, pair, word):
        if state["fresh"]:
            state["merges"].append([pair[0], pair[1]])
            state["fresh"] = False
        return orig(self, pair, word)

    W.NaiveWP._replace_pair = spy
    shim = make_shim()

    def run(corpus, max_vocab):
        m = W.NaiveWP(shim)
        state["merges"], state["fresh"] = [], True
        # initial symbols = the vocabulary before the first merge: train once to max_vocab=0 (no iteration runs)
        m.train(list(corpus), 0)
        initial = sorted(m.vocab)
        state["merges"], state["fresh"] = [], True
        m.train(list(corpus), max_vocab)
        return {"max_vocab": max_vocab, "initial": initial, "merges": state["merges"], "vocab": sorted(m.vocab)}

    rng = random.Random(2929)
    micro = [
        ["This is a sentence.", "Another example sentence."],
        ["aaa aaa"], ["aaaa"], ["aaaaa aaaa aaa aa a"], ["abc abc ab bc"], ["ab ab ab cd cd cd"], ["abab abab"],
        ["x"], [""], ["a b c"], ["hello hello hello world", "hello, world!"], ["ababab ababab bababa"],
        ["abc abd abe abf", "abc abd"], ["zażółć gęślą jaźń", "zażółć gęślą", "jaźń!"], ["a.b.c", "a,b", "(a)"],
        ["low low low low low lower lower newest newest newest newest newest newest widest widest widest"],
        ["aa ab ba bb aa ab ba bb", "aab abb bba baa"], ["mississippi mississippi miss is sip"],
    ]
    for _ in range(14):
        alpha = rng.choice(["ab", "abc", "abcd", "aąb", "xyzż"])
        words = ["".join(rng.choice(alpha) for _ in range(rng.randint(1, 7))) for _ in range(rng.randint(2, 12))]
        micro.append([" ".join(rng.choice(words) for _ in range(rng.randint(3, 14))) for _ in range(rng.randint(1, 4))])
    out = []
    for corpus in micro:
        for mv in (1000,):  # run to exhaustion: every merge of the corpus
            r = run(corpus, mv)
            r["corpus"] = corpus
            out.append(r)
    # a stop in the middle (len(vocab) < max_vocab), on the tutorial corpus of the README
    r = run(micro[0], 25)
    r["corpus"] = micro[0]
    out.append(r)
    # real text: the first sentences of the reference's own corpora (by reference to tests/golden/ref/data)
    for name, n_sent, extra in (("pan_tadeusz", 400, 120), ("train-5K", 5000, 150)):
        corpus = json.load(open(os.path.join(HERE, "ref", "data", name + ".json"), encoding="utf-8"))[:n_sent]
        base = run(corpus, 0)
        r = run(corpus, len(base["initial"]) + extra)
        r["corpus_ref"] = {"file": "ref/data/%s.json" % name, "first": n_sent}
        out.append(r)
        print(name, "initial", len(r["initial"]), "merges", len(r["merges"]), "%.1fs" % (time.time() - t0), flush=True)
    dump("wp_train_order.json", out)
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
