#!/bin/bash
# NOTE: this script measured a form that did not stay in the tree (see profiles/r03_experiments/); its build flag / environment knob exists only in the commit it ran against.
# bpe_lane_kernel: where the time goes, by leaving phases out (-DSWT_LANE_ABL=n builds on the box; their results are wrong)
#   1 no merge rounds (D)   2 + no word list (W)   3 + no pair probes in the split   4 + no split at all (staging, E/F skeleton only)
set -o pipefail
export TMPDIR=/tmp
for v in "" "-DSWT_LANE_ABL=1" "-DSWT_LANE_ABL=2" "-DSWT_LANE_ABL=3" "-DSWT_LANE_ABL=4"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  timeout -k 10 300 python bench.py --workload bpe_encode --corpus open --lean --steps 50 --warmup 5 > gpurun_out/r03p.json 2> gpurun_out/r03p.err || { tail -5 gpurun_out/r03p.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03p.json"))
print("[$v]:", d["value"], "MB/s", d["ms_per_step"], flush=True)
PY
done
