#!/bin/bash
# FastWP wordref: sixteen bytes per lane in the split, against the byte-lane loop (SWT_DD_OLD_SPLIT=1)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "wp_ or config2 or config4 or smoke or cli or metrics or dedup" > gpurun_out/r03k_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03k_pytest.log
if [ $rc -ne 0 ]; then head -60 gpurun_out/r03k_pytest.log; exit $rc; fi
for old in "" 1; do
  export SWT_DD_OLD_SPLIT=$old
  [ -z "$old" ] && unset SWT_DD_OLD_SPLIT
  for w in "wp_encode" "mixed_encode"; do
    n=$(echo $w | tr -d ' -')
    timeout -k 10 400 python bench.py --workload $w > gpurun_out/r03k_$n$old.json 2> gpurun_out/r03k_$n.err; echo "$w rc $?"
    python - <<PY
import json
d=json.load(open("gpurun_out/r03k_$n$old.json"))
r=d.get("roofline") or {}
print("old_split=[$old] $w:", d["value"], "MB/s", d["ms_per_step"], "ms", r.get("dominant_kernel"))
PY
  done
done
SWT_SOAK_SECONDS=90 timeout -k 10 300 python tools/gpu_soak.py > gpurun_out/r03k_soak.txt 2>&1; echo "soak rc $?"; tail -1 gpurun_out/r03k_soak.txt
