"""world_size-2 `gloo` runs on CPU of the corpus-sharded training protocol and of bench.py's rank helpers.
The merges must equal the single-process result of the oracle (= the reference's)."""
import json
import os
import socket

import pytest


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_world(world, corpus, max_vocab, tmp_path):
    import torch.multiprocessing as mp

    from tests import dist_worker

    port = free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=dist_worker.run, args=(r, world, port, corpus, max_vocab, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "rank process failed (exit %s)" % p.exitcode
    return [json.load(open(tmp_path / ("rank%d.json" % r), encoding="utf-8")) for r in range(world)]


@pytest.mark.parametrize("case", ["pan", "ties"])
def test_sharded_training_protocol_world2(case, oracle, corpora, tmp_path, swt):
    pytest.importorskip("torch")
    if case == "pan":
        corpus, max_vocab = corpora["pan"][:120], 160
    else:
        # small alphabet: almost every step is a tie that the global first position must break
        corpus, max_vocab = ["ab ba ab", "ba ab cc", "cc ab ba", "abab baba", "cab bac", "ccc aaa bbb", "abc cba", "bb aa"], 20
    res = run_world(2, corpus, max_vocab, tmp_path)
    ref = oracle.OracleBPETrainer(corpus)
    ref.run(max_vocab)
    want = [list(p) for p in ref.merges_list]
    assert res[0]["merges"] == want and res[1]["merges"] == want
    assert res[0]["vocab"] == res[1]["vocab"] == ref.vocab_size
    assert res[0]["max"] == res[1]["max"] == 2.0
    assert res[0]["sum"] == res[1]["sum"] == float(len(corpus))
    if case == "pan":
        assert res[0]["grown"] >= 1 and res[0]["grown"] == res[1]["grown"]  # the block-overflow path (grow, repeat the exchange) ran


def test_shard_ranges_cover_everything():
    from subword_tokenizers_amd.distributed import shard_range

    for n in (0, 1, 7, 85000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
