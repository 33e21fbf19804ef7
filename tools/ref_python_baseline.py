#!/usr/bin/env python3
"""Time the REFERENCE'S OWN PYTHON on the benchmark corpus (build container only).

The reference never travels to the GPU box, and `NaiveBPE.train` on S85k-open would take hours (about 2 s per merge), so
this script times WINDOWS of the unmodified reference classes, imported with the section 8(c) shim:

  train     FastBPE.train (source/bpe.py:50-112) for `--window` merges at the start, the middle and the end of the
            S85k-open -> vocab 8,000 run.  A window that does not start at merge 0 restarts from a saved symbol state:
            train() always resets and re-reads its corpus (bpe.py:67-81), so the state goes in THROUGH that path -- the
            shim's pre_tokenize_str hands back the words of the state as tuples of symbols (`[s for s in word]` then
            yields the symbols, Counter(words) their frequencies, in first-occurrence order), and max_vocab is the number
            of live symbols + window.  The state itself comes from the C oracle (oracle/swt_oracle.c) run to that merge;
            the merges the reference then makes are compared with the oracle's next merges, which pins the oracle -- and
            through it the device -- on the headline corpus against the real reference.
            Loop time = train(window) - train(0 merges) on the same state (the front end is not the merge loop).
  encode    FastBPE.tokenize (bpe.py:245-249; first 8,000 pretrained merges) on the first 5,000 sentences of S85k-open,
            FastWP.tokenize (wordpiece.py:233-270; V30k) on the first 5,000 sentences of the configs[2] corpus.

Writes profiles/r03_reference_python_baseline.json (timings; bench.py quotes them as cpu_baseline.reference_python,
labelled as measured in the build container) and tests/golden/ref_s85k_open_windows.json (the reference's merges in
the three windows: data for tests/test_oracle_golden.py and the -m gpu parity test of the headline corpus).

Usage: python tools/ref_python_baseline.py [--window 40] [--sample 5000]      (about ten minutes, one core)
"""
import argparse
import json
import os
import signal
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


class _NS:
    pass


class StatePreTokenizer:
    """pre_tokenize_str for a run that restarts from a saved state: sentence "i" -> word i of the state, freq[i] times"""

    def __init__(self, words, freqs):
        self.words, self.freqs = words, freqs

    def pre_tokenize_str(self, key):
        i = int(key)
        return [(self.words[i], (0, 0))] * self.freqs[i]


def shim(pre):
    s = _NS()
    s.backend_tokenizer = _NS()
    s.backend_tokenizer.pre_tokenizer = pre
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=40)
    ap.add_argument("--sample", type=int, default=5000)
    ap.add_argument("--max-vocab", type=int, default=8000)
    args = ap.parse_args()

    import numpy as np
    from tokenizers.pre_tokenizers import BertPreTokenizer

    from oracle import oracle as O
    from subword_tokenizers_amd import synth

    sys.path.insert(0, REF)
    import tqdm as _tqdm  # the progress bar writes to stderr per merge: silence it, it is not the loop
    from source import bpe as ref_bpe
    from source import wordpiece as ref_wp

    class _Quiet:
        def __init__(self, *a, **k):
            pass

        def update(self, n=1):
            pass

        def close(self):
            pass

    ref_bpe.tqdm = _Quiet
    ref_wp.tqdm = _Quiet
    del _tqdm

    sents = synth.s85k_open()
    n_bytes = sum(len(s.encode("utf-8")) for s in sents)
    print("S85k-open: %d sentences, %d bytes" % (len(sents), n_bytes), flush=True)

    # the oracle's whole run: where the windows start, and what the reference must find there
    orc = O.OracleBPETrainer(sents)
    base = orc.vocab_size
    total = args.max_vocab - base
    starts = [0, (total - args.window) // 2, total - args.window]
    print("base symbols %d -> %d merges; windows at %s" % (base, total, starts), flush=True)

    out = {"corpus": "S85k-open (synth.s85k_open(), seed 85000)", "n_sentences": len(sents), "n_bytes": n_bytes,
           "max_vocab": args.max_vocab, "n_base_symbols": base, "n_merges_total": total, "window": args.window,
           "host": "build container: Intel Xeon @ 2.10 GHz, 1 core used (the reference is single-threaded Python)",
           "python": sys.version.split()[0], "windows": []}
    golden = {"corpus": out["corpus"], "max_vocab": args.max_vocab, "window": args.window, "windows": []}

    done = 0
    for start in starts:
        if start > done:  # (max_steps 0 would mean "no limit")
            orc.run(args.max_vocab, start - done)
        done = start
        assert orc.n_merges == start, (orc.n_merges, start)
        n_t = orc.n_symbols
        if start == 0:
            tok_sh = shim(BertPreTokenizer())
            corpus = sents
            live = base
        else:
            syms, woff, freq = orc.export()
            names = {}
            for sid in np.unique(syms).tolist():
                names[sid] = orc.symbol(sid)
            woff = woff.astype(np.int64).tolist()
            slist = syms.tolist()
            words = [tuple(names[x] for x in slist[woff[w]:woff[w + 1]]) for w in range(len(woff) - 1)]
            tok_sh = shim(StatePreTokenizer(words, freq.astype(np.int64).tolist()))
            corpus = [str(i) for i in range(len(words))]
            live = len(names)
        # front end only (zero merges), then front end + window
        t = ref_bpe.FastBPE(tok_sh)
        t0 = time.perf_counter()
        t.train(corpus, live)
        front_s = time.perf_counter() - t0
        assert len(t.merges_list) == 0 and len(t.vocab) == live, (len(t.merges_list), len(t.vocab), live)
        t = ref_bpe.FastBPE(tok_sh)
        t0 = time.perf_counter()
        t.train(corpus, live + args.window)
        full_s = time.perf_counter() - t0
        got = [list(p) for p in t.merges_list]
        # the oracle's next `window` merges
        orc.run(args.max_vocab, args.window)
        done += args.window
        want = [list(p) for p in orc.merges_list[start:start + args.window]]
        assert got == want, "the reference's merges differ from the oracle's in the window at %d:\n%s\n%s" % (start, got[:5], want[:5])
        loop_s = full_s - front_s
        row = {"first_merge": start, "live_symbols_before": int(n_t), "merges": len(got), "front_end_s": round(front_s, 3),
               "front_end_plus_window_s": round(full_s, 3), "loop_s": round(loop_s, 3), "s_per_merge": round(loop_s / len(got), 4)}
        out["windows"].append(row)
        golden["windows"].append({"first_merge": start, "merges": got})
        print("window at %d: N_t %d, %.2f s / merge (front end %.1f s)" % (start, n_t, loop_s / len(got), front_s), flush=True)

    # linear in N_t between the windows (the reference rescans every live symbol each merge): the whole run, extrapolated
    xs = [w["live_symbols_before"] for w in out["windows"]]
    ys = [w["s_per_merge"] for w in out["windows"]]
    seg = [((starts[i + 1] - starts[i]) * (ys[i] + ys[i + 1]) / 2.0) for i in range(len(starts) - 1)]
    whole = sum(seg) + ys[-1] * args.window
    out["train_extrapolated"] = {
        "s_whole_merge_loop": round(whole, 1), "s_per_1k_merges": round(whole / total * 1000, 1), "cores": 1, "kind": "reference",
        "label": "EXTRAPOLATED: trapezoid over the three measured windows (start / middle / end) of the reference's own "
                 "FastBPE.train on S85k-open -> vocab %d; measured in the build container, not on the GPU box" % args.max_vocab,
        "live_symbols_at_windows": xs, "s_per_merge_at_windows": ys}
    print("train, extrapolated: %.0f s for %d merges = %.1f s / 1k merges" % (whole, total, whole / total * 1000), flush=True)

    # encoders
    sub = sents[:args.sample]
    sub_bytes = sum(len(s.encode("utf-8")) for s in sub)
    fb = ref_bpe.FastBPE(shim(BertPreTokenizer()))
    fb.merges_list = [tuple(m) for m in synth.pretrained_merges()[:8000]]
    fb._bpe_ranks = {pair: i for i, pair in enumerate(fb.merges_list)}
    t0 = time.perf_counter()
    toks = [fb.tokenize(s) for s in sub]
    bpe_s = time.perf_counter() - t0
    ob = O.OracleBPE(fb.merges_list)
    assert toks == [ob.tokenize(s) for s in sub], "FastBPE.tokenize differs from the oracle on the S85k-open sample"
    out["bpe_encode"] = {"value": round(sub_bytes / 1e6 / bpe_s, 4), "unit": "MB/s", "cores": 1, "kind": "reference",
                         "sample": "FastBPE.tokenize per sentence, first %d sentences of S85k-open (%.2f MB), first 8,000 pretrained merges; build container" % (len(sub), sub_bytes / 1e6),
                         "tokens": sum(map(len, toks)), "seconds": round(bpe_s, 3)}
    print("FastBPE.tokenize: %.3f MB/s" % (sub_bytes / 1e6 / bpe_s), flush=True)

    vocab = synth.v30k()
    text, off = synth.wp_corpus(args.sample, seed=1000000, vocab=vocab)
    wsents = synth.unpack(text, off, 0, args.sample)
    w_bytes = int(off[args.sample])
    fw = ref_wp.FastWP(shim(BertPreTokenizer()))
    fw.vocab = set(vocab)
    from source.utils import WPTrie_E2E
    t0 = time.perf_counter()
    fw.vocab_trie = WPTrie_E2E(fw.vocab)
    trie_s = time.perf_counter() - t0

    def _alarm(signum, frame):
        raise TimeoutError()

    signal.signal(signal.SIGALRM, _alarm)
    signal.setitimer(signal.ITIMER_REAL, 600)
    t0 = time.perf_counter()
    wtoks = [fw.tokenize(s) for s in wsents]
    wp_s = time.perf_counter() - t0
    signal.setitimer(signal.ITIMER_REAL, 0)
    ow = O.OracleWP(sorted(vocab))
    assert wtoks == [ow.tokenize(s) for s in wsents], "FastWP.tokenize differs from the oracle on the configs[2] sample"
    out["wp_encode"] = {"value": round(w_bytes / 1e6 / wp_s, 4), "unit": "MB/s", "cores": 1, "kind": "reference",
                        "sample": "FastWP.tokenize per sentence, first %d sentences of the configs[2] corpus (%.2f MB), V30k; build container" % (len(wsents), w_bytes / 1e6),
                        "tokens": sum(map(len, wtoks)), "seconds": round(wp_s, 3), "trie_build_s": round(trie_s, 3)}
    print("FastWP.tokenize: %.3f MB/s (trie build %.2f s)" % (w_bytes / 1e6 / wp_s, trie_s), flush=True)

    with open(os.path.join(ROOT, "profiles", "r03_reference_python_baseline.json"), "w") as f:
        json.dump(out, f, indent=1)
    with open(os.path.join(ROOT, "tests", "golden", "ref_s85k_open_windows.json"), "w", encoding="utf-8") as f:
        json.dump(golden, f, ensure_ascii=False, separators=(",", ":"))
    print("wrote profiles/r03_reference_python_baseline.json and tests/golden/ref_s85k_open_windows.json", flush=True)


if __name__ == "__main__":
    main()
