#!/bin/bash
# full -m gpu suite on the tree with the word-lane kernel, the randomized soak, then the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03w_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03w_pytest.log
if [ $rc -ne 0 ]; then head -60 gpurun_out/r03w_pytest.log; exit $rc; fi
SWT_SOAK_SECONDS=150 timeout -k 10 400 python tools/gpu_soak.py > gpurun_out/r03w_soak.txt 2>&1; rc=$?
tail -4 gpurun_out/r03w_soak.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r03w_bench.json 2> gpurun_out/r03w_bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03w_bench.json"))
print(d["value"], d["unit"], d["ms_per_step"], d["roofline"])
for k in ("other_corpus","wp_encode","mixed_encode","train"):
    v=d.get(k)
    if isinstance(v,dict): print(k, {a:v[a] for a in list(v)[:6]})
print(json.dumps(d.get("encode_detail"))[:1500])
PY
