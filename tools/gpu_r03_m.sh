#!/bin/bash
# 32-byte slots with the cached fast read + wordref in two launches, on top of the 16-bytes-per-lane split (one launch for comparison)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "wp_ or bpe_ or config2 or config4 or smoke or cli or metrics or dedup or headline_corpus_encode" > gpurun_out/r03m_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03m_pytest.log
if [ $rc -ne 0 ]; then head -60 gpurun_out/r03m_pytest.log; exit $rc; fi
for one in "" 1; do
  export SWT_DD_ONE_LAUNCH=$one
  [ -z "$one" ] && unset SWT_DD_ONE_LAUNCH
  for w in "wp_encode" "mixed_encode" "bpe_encode --corpus lex"; do
    n=$(echo $w | tr -d ' -')
    timeout -k 10 400 python bench.py --workload $w > gpurun_out/r03m_$n$one.json 2> gpurun_out/r03m_$n.err; echo "$w rc $?"
    python - <<PY
import json
d=json.load(open("gpurun_out/r03m_$n$one.json"))
r=d.get("roofline") or {}
print("one_launch=[$one] $w:", d["value"], "MB/s", d["ms_per_step"], "ms", r.get("dominant_kernel"))
PY
  done
done
