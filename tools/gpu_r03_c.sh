#!/bin/bash
# round 3, third GPU call: the fused WordPiece step (wp_step_kernel) -- parity tests, then the wp_train line fused and generic
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "wp_train or smoke or collision or wordpiece or wp_" > gpurun_out/r03c_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03c_pytest.log
if [ $rc -ne 0 ]; then head -30 gpurun_out/r03c_pytest.log; exit $rc; fi
for g in "" 1; do
  SWT_WP_GENERIC=$g timeout -k 10 200 python bench.py --workload wp_train > gpurun_out/r03c_wp_train_$g.json 2> gpurun_out/r03c_wp_train.err; echo "wp_train generic=[$g] rc $?"
  python -c "import json; d=json.load(open('gpurun_out/r03c_wp_train_$g.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['bytes_model'])"
done
python - <<'PY'
# a larger WordPiece training: train-5K to +1500 merges and S85k-open to +500, device against oracle (every merge)
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
from subword_tokenizers_amd import _native as N, synth, tokenizers
from oracle import oracle as O
N.init(0)
for name, sents, extra in (("train-5K", json.load(open("tests/golden/ref/data/train-5K.json", encoding="utf-8")), 1500), ("S85k-open", synth.s85k_open(), 300)):
    probe = tokenizers.NaiveWP(); probe.train(sents, 0); base = len(probe.vocab); probe.reset()
    N.profile_enable(True); N.profile_read()
    t0 = time.perf_counter()
    tok = tokenizers.NaiveWP(); tok.train(sents, base + extra)
    wall = time.perf_counter() - t0
    ms, _ = N.profile_read(); N.profile_enable(False)
    order = [tuple(m) for m in tok._merge_order]
    st = tok._trainer.stats()
    t0 = time.perf_counter()
    orc = O.OracleWPTrainer(sents); orc.run(base + extra)
    ot = time.perf_counter() - t0
    want = [tuple(m) for m in orc.merges_list]
    bad = next((i for i, (a, b) in enumerate(zip(order, want)) if a != b), None)
    print(name, "merges", len(order), len(want), "first difference", bad, "wall %.3f s, device loop %.2f us/merge, candidates %d, oracle %.1f s" % (wall, ms * 1e3 / max(len(order), 1), st["candidates"], ot), flush=True)
    assert bad is None and len(order) == len(want)
PY
