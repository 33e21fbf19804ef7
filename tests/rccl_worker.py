"""One rank of the world-N RCCL test (tests/test_gpu_multi.py): started as a FRESH process per rank, before anything in it has
touched a GPU.  argv: rank world id_file out_file n_sentences max_vocab.  Rank 0 makes the ncclUniqueId and leaves it in id_file;
every rank opens device `rank`, joins the communicator, trains its contiguous shard of train-5K through the product's sharded
runner (subword_tokenizers_amd.distributed.ShardedBpeTrainer over HipShardEngine + RCCL) and writes the merges it got."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    id_file, out_file = sys.argv[3], sys.argv[4]
    n_sent, max_vocab = int(sys.argv[5]), int(sys.argv[6])
    import numpy as np

    from subword_tokenizers_amd import _native as N
    from subword_tokenizers_amd.distributed import ShardedBpeTrainer

    N.init(rank)
    if rank == 0:
        uid = N.Dist.unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid.tobytes())
        os.replace(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 120:
                raise SystemExit("rank %d: no unique id after 120 s" % rank)
            time.sleep(0.05)
        uid = np.frombuffer(open(id_file, "rb").read(), dtype=np.uint8)
    comm = N.Dist.rccl(rank, world, uid)
    with open(os.path.join(ROOT, "tests", "golden", "ref", "data", "train-5K.json"), encoding="utf-8") as f:
        sents = json.load(f)[:n_sent]
    tr = ShardedBpeTrainer.from_corpus(sents, rank, world, comm)
    try:
        merges = tr.train(max_vocab)
        stats = tr.engine.trainers[0].stats()
    finally:
        tr.engine.close()
        comm.close()
    with open(out_file, "w", encoding="utf-8") as f:
        json.dump({"rank": rank, "merges": [list(m) for m in merges], "vocab": len(tr.vocab), "steps": stats["steps"], "flags": stats["flags"]}, f,
                  ensure_ascii=False)


if __name__ == "__main__":
    main()
