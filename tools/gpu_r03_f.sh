#!/bin/bash
# dedup: wordref in two launches (cached fast reads behind the kernel boundary) against one launch
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "dedup or bpe_ or wp_ or config2 or config4 or headline_corpus_encode or smoke" > gpurun_out/r03f_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03f_pytest.log
if [ $rc -ne 0 ]; then head -40 gpurun_out/r03f_pytest.log; exit $rc; fi
for one in "" 1; do
  export SWT_DD_ONE_LAUNCH=$one
  [ -z "$one" ] && unset SWT_DD_ONE_LAUNCH
  for w in "wp_encode" "mixed_encode" "bpe_encode --corpus lex"; do
    n=$(echo $w | tr -d ' -')
    timeout -k 10 400 python bench.py --workload $w --lean > gpurun_out/r03f_$n.json 2> gpurun_out/r03f_$n.err; echo "$w rc $?"
    python - <<PY
import json
d=json.load(open("gpurun_out/r03f_$n.json"))
print("one_launch=[$one] $w:", d["value"], "MB/s", d["ms_per_step"], "ms")
PY
  done
done
