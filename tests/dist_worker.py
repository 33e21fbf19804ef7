"""Worker for the world_size-2 gloo tests (CPU).  The compute engine here is a small pure-Python stand-in with the
engine interface of subword_tokenizers_amd.distributed (tests may do that; the product's only engine is HIP):
what is under test is the sharding/exchange PROTOCOL -- initial histogram reduction, per-merge delta
all-gather, tie-break by global first position, identical stop decisions on every rank."""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NO_POS = 0xFFFFFFFFFFFFFFFF


class PyEngine:
    def __init__(self, corpus):
        from subword_tokenizers_amd.tokenizers import SubwordTokenizer

        words = Counter(w for sent in SubwordTokenizer().preprocessing(corpus) for w, _ in sent)
        self.words = [[ord(c) for c in w] for w in words]
        self.freq = list(words.values())
        self.start = np.concatenate([[0], np.cumsum([len(w) for w in self.words])]).tolist()
        self.counts = {}
        self.pos_base = 0
        self.pending = self._hist()
        for k, v in self.pending.items():
            self.counts[k] = self.counts.get(k, 0) + v

    def _hist(self, only=None):
        h = {}
        for wi, (w, f) in enumerate(zip(self.words, self.freq)):
            if only is not None and wi not in only:
                continue
            for a, b in zip(w[:-1], w[1:]):
                k = (a << 32) | b
                h[k] = h.get(k, 0) + f
        return h

    def base_symbols(self):
        return np.array(sorted({c for w in self.words for c in w}), dtype=np.int64)

    def set_pos_base(self, base):
        self.pos_base = base

    def take_deltas(self):
        items = [(k, v) for k, v in self.pending.items() if v]
        self.pending = {}
        return (np.array([k for k, _ in items], dtype=np.int64).reshape(-1), np.array([v for _, v in items], dtype=np.int64).reshape(-1))

    def add_remote(self, keys, vals):
        for k, v in zip(np.asarray(keys).tolist(), np.asarray(vals).tolist()):
            self.counts[k] = self.counts.get(k, 0) + v

    def best(self):
        live = {k: v for k, v in self.counts.items() if v > 0}
        if not live:
            return 0, 0, 0, 0, NO_POS
        mx = max(live.values())
        tied = [k for k, v in live.items() if v == mx]
        if len(tied) == 1:
            return tied[0] >> 32, tied[0] & 0xFFFFFFFF, mx, 1, NO_POS
        cand = set(tied)
        for wi, w in enumerate(self.words):
            for i, (a, b) in enumerate(zip(w[:-1], w[1:])):
                if ((a << 32) | b) in cand:
                    return a, b, mx, len(tied), self.pos_base + self.start[wi] + i
        return 0xFFFFFFFF, 0xFFFFFFFF, mx, len(tied), NO_POS

    def apply(self, left, right, merged):
        touched = {wi for wi, w in enumerate(self.words) if any(a == left and b == right for a, b in zip(w[:-1], w[1:]))}
        before = self._hist(touched)
        for wi in touched:
            w, out, i = self.words[wi], [], 0
            while i < len(w):
                if i + 1 < len(w) and w[i] == left and w[i + 1] == right:
                    out.append(merged)
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            self.words[wi] = out
        after = self._hist(touched)
        delta = {k: after.get(k, 0) - before.get(k, 0) for k in set(before) | set(after)}
        self.pending = {k: v for k, v in delta.items() if v}
        for k, v in self.pending.items():
            self.counts[k] = self.counts.get(k, 0) + v


def run(rank, world, port, corpus, max_vocab, out_dir):
    import torch.distributed as dist

    from subword_tokenizers_amd.distributed import ShardedBpeTrainer, TorchGroup, reduce_scalar, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(len(corpus), rank, world)
        assert ShardedBpeTrainer.shard(corpus, rank, world) == corpus[lo:hi]
        tr = ShardedBpeTrainer(PyEngine(corpus[lo:hi]), rank, world, TorchGroup(dist, "cpu"))
        merges = tr.train(max_vocab)
        mx = reduce_scalar(dist, float(rank + 1), "max", "cpu")
        sm = reduce_scalar(dist, float(hi - lo), "sum", "cpu")
        import json

        with open(os.path.join(out_dir, "rank%d.json" % rank), "w", encoding="utf-8") as f:
            json.dump({"merges": merges, "vocab": len(tr.vocab), "max": mx, "sum": sm}, f, ensure_ascii=False)
    finally:
        dist.destroy_process_group()
