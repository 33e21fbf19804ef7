#!/bin/bash
# round 3: dedup slots with an inline 16-byte prefix + table sized from the last call -- full -m gpu suite, then the encode lines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03e_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03e_pytest.log
if [ $rc -ne 0 ]; then head -40 gpurun_out/r03e_pytest.log; exit $rc; fi
for w in "wp_encode" "mixed_encode" "bpe_encode --corpus lex" "bpe_encode --corpus open"; do
  n=$(echo $w | tr -d ' -')
  timeout -k 10 400 python bench.py --workload $w > gpurun_out/r03e_$n.json 2> gpurun_out/r03e_$n.err; echo "$w rc $?"
  python - <<PY
import json
d=json.load(open("gpurun_out/r03e_$n.json"))
r=d.get("roofline") or {}
print("$w:", d["value"], "MB/s", d["ms_per_step"], "ms; call us", r.get("kernel_us"), "dominant", r.get("dominant_kernel"), "e2e", d.get("end_to_end_mb_s"), (d.get("encode_detail") or {}))
PY
done
