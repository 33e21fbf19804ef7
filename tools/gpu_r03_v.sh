#!/bin/bash
# NOTE: this script measured a form that did not stay in the tree (see profiles/r03_experiments/); its build flag / environment knob exists only in the commit it ran against.
# bpe_lane_group_kernel: tiles per workgroup (rebuilds on the box)
set -o pipefail
export TMPDIR=/tmp
for v in "-DSWT_LANE_GROUP=2" "-DSWT_LANE_GROUP=2 -DSWT_LANE_TILE=256 -DSWT_LANE_CAP=384" "-DSWT_LANE_GROUP=8 -DSWT_LANE_TILE=192 -DSWT_LANE_CAP=256" "-DSWT_LANE_GROUP=4 -DSWT_LANE_TILE=256 -DSWT_LANE_CAP=384"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  timeout -k 10 300 python bench.py --workload bpe_encode --corpus open --lean --steps 50 --warmup 5 > gpurun_out/r03v.json 2> gpurun_out/r03v.err || { tail -5 gpurun_out/r03v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03v.json"))
print("[$v]:", d["value"], "MB/s", d["ms_per_step"], flush=True)
PY
done
