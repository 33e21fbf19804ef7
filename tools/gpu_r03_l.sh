#!/bin/bash
# both wordref modes with the 16-bytes-per-lane split: full suite, A/B of the encode lines, soak
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03l_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03l_pytest.log
if [ $rc -ne 0 ]; then head -60 gpurun_out/r03l_pytest.log; exit $rc; fi
for old in "" 1; do
  export SWT_DD_OLD_SPLIT=$old
  [ -z "$old" ] && unset SWT_DD_OLD_SPLIT
  for w in "wp_encode" "mixed_encode" "bpe_encode --corpus lex"; do
    n=$(echo $w | tr -d ' -')
    timeout -k 10 400 python bench.py --workload $w --lean > gpurun_out/r03l_$n$old.json 2> gpurun_out/r03l_$n.err; echo "$w rc $?"
    python - <<PY
import json
d=json.load(open("gpurun_out/r03l_$n$old.json"))
print("old_split=[$old] $w:", d["value"], "MB/s", d["ms_per_step"], "ms")
PY
  done
done
SWT_SOAK_SECONDS=150 timeout -k 10 300 python tools/gpu_soak.py > gpurun_out/r03l_soak.txt 2>&1; echo "soak rc $?"; tail -1 gpurun_out/r03l_soak.txt
