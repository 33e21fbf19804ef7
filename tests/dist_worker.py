"""Worker for the world_size-2 gloo tests (CPU).  The engine here is a small pure-Python stand-in with the two-method engine
interface of subword_tokenizers_amd.distributed (tests may do that; the product's only engine is HIP + RCCL).  It runs the
SAME exchange protocol as csrc/swt_dist.hip over torch.distributed collectives:
    begin()  all-gather of the initial symbols; one-off all-gather of the local histograms, every rank adds the others'
    per merge  all-gather of one (first position, pair) line per rank for the tie-break (the first rank that holds a tied
               pair wins), local apply, all-gather of one fixed-size block of (pair, delta) records per rank with a header
               (count, overflow), every rank adds EVERY block -- its own too -- to its replica
What is under test is the host loop of ShardedBpeTrainer.train on top of it: the same stop decisions, interning and merges
on every rank, equal to the single-process result."""
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NO_POS = 0xFFFFFFFFFFFFFFFF
BLOCK = 12  # records per block: small on purpose, so that the overflow path (grow + repeat the exchange) runs


class PyShardEngine:
    def __init__(self, corpus, rank, world, dist):
        import torch

        from subword_tokenizers_amd.tokenizers import SubwordTokenizer

        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        words = Counter(w for sent in SubwordTokenizer().preprocessing(corpus) for w, _ in sent)
        self.words = [[ord(c) for c in w] for w in words]
        self.freq = list(words.values())
        self.counts = {}
        self.block = BLOCK
        self.grown = 0

    # ---- collectives
    def _gather(self, arr):
        t = self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64))
        out = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.numpy() for o in out]

    def _gather_var(self, arr):
        sizes = [int(a[0]) for a in self._gather(np.array([len(arr)], dtype=np.int64))]
        pad = np.zeros(max(sizes + [1]), dtype=np.int64)
        pad[:len(arr)] = arr
        return [g[:n] for g, n in zip(self._gather(pad), sizes)]

    # ---- local state
    def _hist(self, only=None):
        h = {}
        for wi, (w, f) in enumerate(zip(self.words, self.freq)):
            if only is not None and wi not in only:
                continue
            for a, b in zip(w[:-1], w[1:]):
                k = (a << 32) | b
                h[k] = h.get(k, 0) + f
        return h

    def begin(self):
        base = self._gather_var(np.array(sorted({c for w in self.words for c in w}), dtype=np.int64))
        mine = self._hist()
        keys = self._gather_var(np.array(list(mine.keys()), dtype=np.int64))
        vals = self._gather_var(np.array(list(mine.values()), dtype=np.int64))
        for ks, vs in zip(keys, vals):
            for k, v in zip(ks.tolist(), vs.tolist()):
                self.counts[k] = self.counts.get(k, 0) + v
        return np.array(sorted({int(c) for b in base for c in b}), dtype=np.uint32)

    def _local_first(self, tied):
        for wi, w in enumerate(self.words):
            for i, (a, b) in enumerate(zip(w[:-1], w[1:])):
                if ((a << 32) | b) in tied:
                    return (wi << 32) | i, (a << 32) | b
        return -1, -1

    def _apply(self, left, right, merged):
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
        return {k: after.get(k, 0) - before.get(k, 0) for k in set(before) | set(after) if after.get(k, 0) != before.get(k, 0)}

    def _exchange(self, delta):
        while True:
            blk = np.zeros(2 * self.block, dtype=np.int64)
            blk[0], blk[1] = len(delta), int(len(delta) + 1 > self.block)
            if not blk[1]:
                for j, (k, v) in enumerate(delta.items()):
                    blk[2 + 2 * j], blk[3 + 2 * j] = k, v
            blocks = self._gather(blk)
            if not any(int(b[1]) for b in blocks):
                break
            self.block = 2 * (max(int(b[0]) for b in blocks) + 2)  # nothing was added anywhere: bigger blocks, again
            self.grown += 1
        for b in blocks:
            for j in range(int(b[0])):
                k, v = int(b[2 + 2 * j]), int(b[3 + 2 * j])
                self.counts[k] = self.counts.get(k, 0) + v

    def run(self, max_steps, first_merged):
        lefts, rights, counts = [], [], []
        for step in range(max_steps):
            live = {k: v for k, v in self.counts.items() if v > 0}
            if not live:
                break
            mx = max(live.values())
            tied = {k for k, v in live.items() if v == mx}
            pos, key = self._local_first(tied) if len(tied) > 1 else (-1, -1)
            lines = self._gather(np.array([pos, key], dtype=np.int64))
            if len(tied) > 1:
                key = next(int(l[1]) for l in lines if int(l[0]) >= 0)
            else:
                key = next(iter(tied))
            left, right = key >> 32, key & 0xFFFFFFFF
            self._exchange(self._apply(left, right, first_merged + step))
            lefts.append(left)
            rights.append(right)
            counts.append(mx)
        return np.array(lefts, dtype=np.uint32), np.array(rights, dtype=np.uint32), np.array(counts, dtype=np.uint64)


def run(rank, world, port, corpus, max_vocab, out_dir):
    import torch.distributed as dist

    from subword_tokenizers_amd.distributed import ShardedBpeTrainer, reduce_scalar, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(len(corpus), rank, world)
        assert ShardedBpeTrainer.shard(corpus, rank, world) == corpus[lo:hi]
        eng = PyShardEngine(corpus[lo:hi], rank, world, dist)
        tr = ShardedBpeTrainer(eng, rank, world)
        merges = tr.train(max_vocab)
        mx = reduce_scalar(dist, float(rank + 1), "max", "cpu")
        sm = reduce_scalar(dist, float(hi - lo), "sum", "cpu")
        import json

        with open(os.path.join(out_dir, "rank%d.json" % rank), "w", encoding="utf-8") as f:
            json.dump({"merges": merges, "vocab": len(tr.vocab), "max": mx, "sum": sm, "grown": eng.grown}, f, ensure_ascii=False)
    finally:
        dist.destroy_process_group()
