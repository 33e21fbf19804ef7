"""Corpus-sharded BPE training: one process per GPU, the pair histogram reduced over RCCL each merge.

The reference has no parallelism at all; this is the multi-GPU form of its merge loop
(source/bpe.py:88-111, citations relative to /root/reference) that the north star asks for.  Every rank owns a
contiguous range of sentences, pre-tokenizes and dedups it locally, and keeps

    * its own symbol stream (only local words), and
    * the pair histogram of the WHOLE corpus, replicated.

The histogram is a hash table whose slot assignment is rank-local, so a dense element-wise all-reduce cannot be
used on it.  It is reduced sparsely instead: each rank publishes the (pair, delta) list its last merge
produced (the whole local histogram the first time), the lists are all-gathered over RCCL, and every rank adds
the other ranks' lists to its replica -- the sum is the all-reduce of the histogram, in KB per merge instead of
a table-sized message.  All replicas then hold the same counts, so every rank derives the same maximum and
the same tie set on its own; only a tie needs one more tiny all-gather: (first position, pair) per rank, minimum
wins (source/bpe.py:102: the first-inserted maximum of the Counter = the earliest (word, position) in scan
order; ranks are ordered by their sentence ranges, so pos_base = rank << 40 keeps that order global).

Correct-by-construction pieces:
  ShardedBpeTrainer.steps()   the protocol as a generator that yields at every collective
  TorchGroup                  runs it over torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU)
  LocalGroup                  runs several shards in lockstep inside one process (single-GPU parity test)
The compute engine is the HIP trainer (`_native.BpeTrainer`); tests may inject another engine with the same five
methods to exercise the protocol on CPU (the product never does).
"""
from typing import List, Optional, Tuple

import numpy as np

from . import _native as N
from .tokenizers import _SymbolTable

POS_SHIFT = 40
NO_POS = N.NO_POS


class HipEngine:
    """Adapter of the device trainer to the five-method engine interface, with torch-allocated exchange buffers."""

    def __init__(self, trainer: "N.BpeTrainer", torch_mod):
        self.t = trainer
        self.torch = torch_mod
        self._cap = 1 << 16
        self._keys = self._vals = None

    @classmethod
    def from_corpus(cls, corpus: List[str]):
        import torch

        text, off = N.pack_utf8([s.lower() for s in corpus])
        return cls(N.BpeTrainer.from_text(text, off), torch)

    def base_symbols(self) -> np.ndarray:
        return self.t.base_symbols()

    def set_pos_base(self, base: int) -> None:
        self.t.set_pos_base(base)

    def _buffers(self):
        if self._keys is None or self._keys.numel() < self._cap:
            self._keys = self.torch.empty(self._cap, dtype=self.torch.int64, device="cuda")
            self._vals = self.torch.empty(self._cap, dtype=self.torch.int64, device="cuda")
        return self._keys, self._vals

    def take_deltas(self):
        """-> (keys int64 tensor, deltas int64 tensor) on the device: what the last apply changed"""
        stream = self.torch.cuda.current_stream().cuda_stream
        while True:
            k, v = self._buffers()
            try:
                n = self.t.take_deltas(k.data_ptr(), v.data_ptr(), k.numel(), stream)
                break
            except N.SwtError as e:
                if e.code != N.ERR_CAPACITY or self._cap > (1 << 34):
                    raise
                self._cap *= 4
        self.torch.cuda.current_stream().synchronize()
        return k[:n], v[:n]

    def add_remote(self, keys, vals) -> None:
        if keys.numel():
            keys = keys.contiguous()
            vals = vals.contiguous()
            self.t.add_remote(keys.data_ptr(), vals.data_ptr(), keys.numel(), self.torch.cuda.current_stream().cuda_stream)
            self.torch.cuda.current_stream().synchronize()  # keys/vals may be freed by the caller

    def best(self):
        return self.t.best()

    def apply(self, left, right, merged) -> None:
        self.t.apply(left, right, merged)


class ShardedBpeTrainer:
    """One rank of corpus-sharded NaiveBPE.train / FastBPE.train."""

    def __init__(self, engine, rank: int, world: int, group=None):
        self.engine, self.rank, self.world, self.group = engine, rank, world, group
        self.merges_list: List[Tuple[str, str]] = []
        self.vocab: set = set()
        engine.set_pos_base(rank << POS_SHIFT)

    @staticmethod
    def shard(corpus: List[str], rank: int, world: int) -> List[str]:
        n = len(corpus)
        return corpus[rank * n // world:(rank + 1) * n // world]

    @classmethod
    def from_corpus(cls, corpus: List[str], rank: int, world: int, group=None):
        return cls(HipEngine.from_corpus(cls.shard(corpus, rank, world)), rank, world, group)

    def _exchange_deltas(self):
        keys, vals = self.engine.take_deltas()
        gathered = yield ("gather_pairs", keys, vals)
        for r, (k, v) in enumerate(gathered):
            if r != self.rank:
                self.engine.add_remote(k, v)

    def steps(self, max_vocab: int):
        """The training protocol; yields ("gather_*", payload...) at every collective and expects the list of all
        ranks' payloads back.  bpe.py:88-111."""
        syms = _SymbolTable()
        done = set()
        base = yield ("gather_array", np.asarray(self.engine.base_symbols(), dtype=np.int64))
        self.vocab = {chr(int(c)) for arr in base for c in arr}  # bpe.py:75 over the whole corpus
        yield from self._exchange_deltas()  # the one-off reduction of the local histograms
        while len(self.vocab) < max_vocab:  # bpe.py:88
            left, right, count, tied, pos = self.engine.best()  # global counts: identical on every rank
            if count == 0:  # bpe.py:98-99
                break
            if tied > 1:  # bpe.py:102 tie: earliest (word, position) over all shards
                cands = yield ("gather_array", np.array([pos if pos != NO_POS else -1, left, right], dtype=np.int64))
                live = [c for c in cands if c[0] >= 0]
                win = min(live, key=lambda c: int(c[0]))
                left, right = int(win[1]), int(win[2])
            ls, rs = syms.string(left), syms.string(right)
            if (left, right) in done:
                # symbols only ever merge, so a fully merged pair cannot come back: the replicas have diverged
                raise RuntimeError("pair histogram inconsistent: %r selected twice" % ((ls, rs),))
            done.add((left, right))
            self.vocab.add(ls + rs)  # bpe.py:103
            self.merges_list.append((ls, rs))  # bpe.py:104
            self.engine.apply(left, right, syms.intern(ls + rs))  # bpe.py:108-111
            yield from self._exchange_deltas()
        return self.merges_list

    def train(self, max_vocab: int) -> List[Tuple[str, str]]:
        """Run the protocol over this rank's process group (torch.distributed)."""
        if self.group is None:
            raise ValueError("no process group")
        return self.group.run(self.steps(max_vocab))


class TorchGroup:
    """Collectives of the protocol over torch.distributed: "nccl" (RCCL over xGMI) on GPUs, "gloo" on CPU."""

    def __init__(self, dist, device):
        self.dist, self.device = dist, device
        import torch

        self.torch = torch

    def _sizes(self, n: int) -> List[int]:
        t = self.torch.tensor([n], dtype=self.torch.int64, device=self.device)
        out = [self.torch.zeros_like(t) for _ in range(self.dist.get_world_size())]
        self.dist.all_gather(out, t)
        return [int(x.item()) for x in out]

    def _gather_var(self, t):
        sizes = self._sizes(int(t.numel()))
        m = max(sizes + [1])
        pad = self.torch.zeros(m, dtype=self.torch.int64, device=self.device)
        pad[:t.numel()] = t
        out = [self.torch.empty_like(pad) for _ in sizes]
        self.dist.all_gather(out, pad)
        return [o[:s] for o, s in zip(out, sizes)]

    def collective(self, op):
        kind = op[0]
        if kind == "gather_array":
            t = self.torch.from_numpy(np.ascontiguousarray(op[1], dtype=np.int64)).to(self.device)
            return [g.cpu().numpy() for g in self._gather_var(t)]
        if kind == "gather_pairs":
            keys = op[1].to(self.device) if hasattr(op[1], "to") else self.torch.from_numpy(op[1]).to(self.device)
            vals = op[2].to(self.device) if hasattr(op[2], "to") else self.torch.from_numpy(op[2]).to(self.device)
            return list(zip(self._gather_var(keys), self._gather_var(vals)))
        raise ValueError(kind)

    def run(self, gen):
        try:
            op = next(gen)
            while True:
                op = gen.send(self.collective(op))
        except StopIteration as stop:
            return stop.value


class LocalGroup:
    """Several shards of one job inside ONE process, advanced in lockstep (single-GPU test of the protocol)."""

    def __init__(self, world: int):
        self.world = world

    @staticmethod
    def run_lockstep(shards: List[ShardedBpeTrainer], max_vocab: int):
        gens = [s.steps(max_vocab) for s in shards]
        ops = [next(g) for g in gens]
        results: List[Optional[list]] = [None] * len(gens)
        while True:
            kinds = {op[0] for op in ops}
            assert len(kinds) == 1, "shards diverged: %s" % kinds
            # snapshot every rank's contribution first: a shard that is resumed earlier reuses its exchange buffers
            if ops[0][0] == "gather_array":
                payload = [np.array(op[1], copy=True) for op in ops]
            else:
                payload = [(op[1].clone(), op[2].clone()) for op in ops]
            nxt = []
            done = 0
            for i, g in enumerate(gens):
                try:
                    nxt.append(g.send(payload))
                except StopIteration as stop:
                    results[i] = stop.value
                    nxt.append(None)
                    done += 1
            if done:
                assert done == len(gens), "shards finished at different steps"
                break
            ops = nxt
        assert all(r == results[0] for r in results)
        return results[0]


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


def train_sharded(corpus: List[str], max_vocab: int, rank: int, world: int, dist):
    """bench.py's N > 1 training leg: this rank's shard of `corpus` through ShardedBpeTrainer over torch.distributed
    (backend "nccl" = RCCL).  -> (merges_list, trainer info of this rank)"""
    tr = ShardedBpeTrainer.from_corpus(corpus, rank, world, TorchGroup(dist, "cuda"))
    merges = [tuple(m) for m in tr.train(max_vocab)]
    info = tr.engine.t.info()
    tr.engine.t.close()
    return merges, info
