"""Corpus-sharded BPE training: one process per GPU, the pair histogram reduced over RCCL every merge.

The reference has no parallelism at all; this is the multi-GPU form of its merge loop
(source/bpe.py:88-111, citations relative to /root/reference) that the north star asks for.  Every rank owns a
contiguous range of sentences, pre-tokenizes and dedups it locally, and keeps

    * its own symbol stream and inverted index (only local words), and
    * the pair histogram of the WHOLE corpus, replicated.

The merge loop itself runs in C++ (csrc/swt_dist.hip, `swt_bpe_train_run_sharded`): per merge ONE fixed-size
ncclAllGather of packed (pair, delta) records -- every rank adds every block to its replica, so all replicas hold the
same counts and derive the same maximum and tie set -- and ONE 16-byte all-gather of (first position, pair) for the
tie-break (source/bpe.py:102: the first-inserted maximum of the Counter = the earliest (word, position) in scan order;
ranks are ordered by their sentence ranges, so the first rank holding a tied pair wins).  Both are enqueued on the
training stream; up to 256 merges go out per call and nothing passes through Python in between.

What stays in Python is what the reference does in Python: the string set, the stop test `len(vocab) < max_vocab`
(bpe.py:88,103) and the merges list -- `ShardedBpeTrainer.train`, identical on every rank because the device outputs are.

  ShardedBpeTrainer   one rank of NaiveBPE.train / FastBPE.train over an engine with two methods:
                        begin() -> distinct initial symbols of the whole corpus;  run(k, first_id) -> (left, right, count)
  HipShardEngine      the product engine: _native.BpeTrainer + _native.Dist (RCCL, or the one-process loop-back)
  rccl_dist()         communicator over an initialised torch.distributed group (the 128-byte id is broadcast from rank 0)
Tests may inject another engine with the same two methods to exercise the host loop on CPU (tests/dist_worker.py runs the
same exchange protocol in pure Python over gloo); the product never does.
"""
from typing import List, Tuple

import numpy as np

from . import _native as N
from .tokenizers import _SymbolTable


class HipShardEngine:
    """This rank's device trainer + the communicator.  `local` engines (loop-back) hold every rank's trainer."""

    def __init__(self, trainers: List["N.BpeTrainer"], dist: "N.Dist"):
        self.trainers, self.dist = trainers, dist

    @classmethod
    def from_corpus(cls, shard: List[str], dist: "N.Dist"):
        text, off = N.pack_and_lower(shard)
        return cls([N.BpeTrainer.from_text(text, off)], dist)

    def begin(self) -> np.ndarray:
        return self.dist.shard_begin(self.trainers)

    def run(self, max_steps: int, first_merged: int):
        return self.dist.run(self.trainers, max_steps, first_merged)

    def info(self):
        return self.trainers[0].info()

    def close(self):
        for t in self.trainers:
            t.close()
        self.trainers = []


class ShardedBpeTrainer:
    """One rank of corpus-sharded NaiveBPE.train / FastBPE.train (bpe.py:50-112)."""

    def __init__(self, engine, rank: int, world: int):
        self.engine, self.rank, self.world = engine, rank, world
        self.merges_list: List[Tuple[str, str]] = []
        self.vocab: set = set()

    @staticmethod
    def shard(corpus: List[str], rank: int, world: int) -> List[str]:
        lo, hi = shard_range(len(corpus), rank, world)
        return corpus[lo:hi]

    @classmethod
    def from_corpus(cls, corpus: List[str], rank: int, world: int, dist: "N.Dist"):
        return cls(HipShardEngine.from_corpus(cls.shard(corpus, rank, world), dist), rank, world)

    def train(self, max_vocab: int) -> List[Tuple[str, str]]:
        syms = _SymbolTable()
        done = set()
        self.vocab = {chr(int(c)) for c in self.engine.begin()}  # bpe.py:75 over the whole corpus
        exhausted = False
        while len(self.vocab) < max_vocab and not exhausted:  # bpe.py:88
            want = max_vocab - len(self.vocab)
            first = N.SYM_BASE + len(syms.strings)
            lefts, rights, _counts = self.engine.run(want, first)
            if len(lefts) < want:
                exhausted = True  # bpe.py:98-99: no pair left in any shard
            for i in range(len(lefts)):
                left, right = int(lefts[i]), int(rights[i])
                if (left, right) in done:
                    # symbols only ever merge, so a fully merged pair cannot come back: the replicas have diverged
                    raise RuntimeError("pair histogram inconsistent: %r selected twice" % ((left, right),))
                done.add((left, right))
                ls, rs = syms.string(left), syms.string(right)
                merged = syms.intern(ls + rs)
                self.vocab.add(ls + rs)  # bpe.py:103
                self.merges_list.append((ls, rs))  # bpe.py:104
                if merged != first + i:
                    # two merges spelling one string (never observed): the unsharded trainer replays from
                    # the text; a sharded run has no replay, so say so instead of diverging from the reference
                    raise RuntimeError("merged string %r already names symbol %d: sharded training cannot replay the collision"
                                       % (ls + rs, merged))
        return self.merges_list


def rccl_dist(dist, rank: int, world: int, device="cuda") -> "N.Dist":
    """An RCCL communicator for the C++ runner over an initialised torch.distributed group: rank 0 makes the 128-byte id
    (ncclGetUniqueId), torch.distributed broadcasts it."""
    import torch

    if dist is None or world == 1:
        return N.Dist.rccl(0, 1, N.Dist.unique_id())
    uid = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        uid.copy_(torch.from_numpy(N.Dist.unique_id()))
    dist.broadcast(uid, src=0)
    return N.Dist.rccl(rank, world, uid.cpu().numpy())


def train_sharded(corpus: List[str], max_vocab: int, rank: int, world: int, dist):
    """bench.py's N > 1 training leg: this rank's shard of `corpus` through the C++ runner over RCCL.
    -> (merges_list, trainer info of this rank)"""
    comm = rccl_dist(dist, rank, world)
    tr = ShardedBpeTrainer.from_corpus(corpus, rank, world, comm)
    try:
        merges = [tuple(m) for m in tr.train(max_vocab)]
        info = dict(tr.engine.info())
        info.update(tr.engine.trainers[0].stats())  # bench.py's bytes model reads entries_scanned
    finally:
        tr.engine.close()
        comm.close()
    return merges, info


def train_sharded_loopback(corpus: List[str], max_vocab: int, world: int):
    """All `world` shards as trainers of THIS process (one GPU): the same C++ runner with device copies for collectives."""
    comm = N.Dist.loopback(world)
    trainers = []
    for r in range(world):
        text, off = N.pack_and_lower(ShardedBpeTrainer.shard(corpus, r, world))
        trainers.append(N.BpeTrainer.from_text(text, off))
    tr = ShardedBpeTrainer(HipShardEngine(trainers, comm), 0, world)
    try:
        return [tuple(m) for m in tr.train(max_vocab)], [t.stats() for t in trainers]
    finally:
        tr.engine.close()
        comm.close()


# ---- helpers shared with bench.py (device-agnostic so the gloo tests cover them) ------------------------

def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of `n_items` sentences for `rank` (encode shards by sentence; no collective)."""
    return rank * n_items // world, (rank + 1) * n_items // world


def reduce_scalar(dist, value: float, op: str, device):
    """MAX / SUM of one float over the ranks (identity when there is no process group)."""
    if dist is None:
        return value
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}[op])
    return float(t.item())
