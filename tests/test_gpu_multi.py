"""World-2 run of the sharded trainer over RCCL between two PROCESSES (one per GPU): what bench.py --gpus N does for its `train`
block.  Needs two GPUs: on the one-GPU box of the round-end tests it skips (the loop-back and world-1 RCCL tests of
test_gpu_parity.py / test_gpu_configs.py cover the same kernels and the same exchange protocol there).  Every rank is a fresh
child process started BEFORE anything here touches a GPU (the parent only counts devices)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_count():
    import torch

    return torch.cuda.device_count()  # does not initialise the GPU on this image


@pytest.mark.parametrize("generic", ["0", "1"])
def test_sharded_training_over_rccl_world2(oracle, corpora, tmp_path, generic):
    if _device_count() < 2:
        pytest.skip("needs two GPUs (one process per GPU); the one-GPU box runs the loop-back and world-1 RCCL forms")
    n_sent, extra = 3000, 400
    sents = corpora["t5k"][:n_sent]
    ref = oracle.OracleBPETrainer(sents)
    target = ref.vocab_size + extra
    ref.run(target)
    want = [list(p) for p in ref.merges_list]
    id_file = str(tmp_path / "nccl_id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SWT_DIST_GENERIC=generic)
    procs = []
    for r in range(2):
        out = str(tmp_path / ("rank%d.json" % r))
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), str(r), "2", id_file, out, str(n_sent),
                                             str(target)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    results = []
    for out, p in procs:
        try:
            log, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for _o, q in procs:
                q.kill()  # the exact children started above
            pytest.fail("a rank did not finish within 300 s")
        assert p.returncode == 0, log[-2000:]
        with open(out, encoding="utf-8") as f:
            results.append(json.load(f))
    for res in results:
        assert res["merges"] == want, res["rank"]
        assert res["vocab"] == target and res["flags"] & 1 == 0
    if generic == "0":
        assert results[0]["steps"] < extra  # tied merges were batched
