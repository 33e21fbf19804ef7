"""Sharded BPE training on ONE GPU: S85k-open -> vocab 8,000 through the sharded runner of csrc/swt_dist.hip,
  * over an RCCL communicator of one rank (the collectives are real ncclAllGather calls on the training stream), and
  * over the loop-back communicator with 2 and 4 shards (all ranks are trainers of this process, device copies for collectives),
each in the fast form (several tied merges per step) and in round 2's generic form (SWT_DIST_GENERIC=1), beside the unsharded
FastBPE.train.  Prints one JSON line per run: wall per merge, device loop per merge (HIP events around every round trip), steps.
No oracle here (tests/test_gpu_configs.py::test_headline_corpus_sharded_fast_path holds the parity); the merges of every run
are compared with the unsharded run's."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from subword_tokenizers_amd import _native as N
from subword_tokenizers_amd import synth, tokenizers
from subword_tokenizers_amd.distributed import HipShardEngine, ShardedBpeTrainer, train_sharded_loopback

N.init(0)
sents = synth.s85k_open()


def timed(fn, repeats=2):
    out = None
    best = None
    for _ in range(1 + repeats):
        N.profile_enable(True)
        N.profile_read()
        t0 = time.perf_counter()
        out = fn()
        wall = time.perf_counter() - t0
        ms, _n = N.profile_read()
        if best is None or wall < best[0]:
            best = (wall, ms / 1e3)
    N.profile_enable(False)
    return out, best


def unsharded():
    tok = tokenizers.FastBPE()
    tok.train(sents, 8000)
    m = [tuple(x) for x in tok.merges_list]
    st = tok._trainer.stats()
    tok.reset()
    return m, st


def rccl_world1():
    comm = N.Dist.rccl(0, 1, N.Dist.unique_id())
    tr = ShardedBpeTrainer.from_corpus(sents, 0, 1, comm)
    try:
        m = [tuple(x) for x in tr.train(8000)]
        st = tr.engine.trainers[0].stats()
    finally:
        tr.engine.close()
        comm.close()
    return m, st


(base, st0), (wall, loop) = timed(unsharded)
print(json.dumps({"run": "unsharded FastBPE.train", "merges": len(base), "steps": st0["steps"], "us_per_merge_wall": round(wall / len(base) * 1e6, 2),
                  "us_per_merge_device_loop": round(loop / len(base) * 1e6, 2)}), flush=True)
for generic in ("0", "1"):
    os.environ["SWT_DIST_GENERIC"] = generic
    form = "generic (one merge per step)" if generic == "1" else "fast (tied merges batched)"
    (m, st), (wall, loop) = timed(rccl_world1, repeats=1)
    assert m == base, "RCCL world-1 merges differ from the unsharded run"
    print(json.dumps({"run": "RCCL world 1, " + form, "merges": len(m), "steps": st["steps"], "us_per_merge_wall": round(wall / len(m) * 1e6, 2),
                      "us_per_merge_device_loop": round(loop / len(m) * 1e6, 2)}), flush=True)
    for world in (2, 4):
        (res, (wall, loop)) = timed(lambda: train_sharded_loopback(sents, 8000, world), repeats=1)
        m, stats = res
        assert m == base, "loop-back merges differ from the unsharded run (world %d)" % world
        print(json.dumps({"run": "loop-back %d shards, %s" % (world, form), "merges": len(m), "steps": stats[0]["steps"],
                          "us_per_merge_wall": round(wall / len(m) * 1e6, 2), "us_per_merge_device_loop": round(loop / len(m) * 1e6, 2)}), flush=True)
