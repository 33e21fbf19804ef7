#!/bin/bash
# bpe_lane_kernel: tile against chunk size (how many tiles need a second chunk) -- rebuilds on the box
set -o pipefail
export TMPDIR=/tmp
for v in "-DSWT_LANE_TILE=384 -DSWT_LANE_CAP=512" "-DSWT_LANE_TILE=320 -DSWT_LANE_CAP=512" "-DSWT_LANE_TILE=352 -DSWT_LANE_CAP=512" "-DSWT_LANE_TILE=384 -DSWT_LANE_CAP=576" "-DSWT_LANE_TILE=384 -DSWT_LANE_CAP=640" "-DSWT_LANE_TILE=448 -DSWT_LANE_CAP=640" "-DSWT_LANE_TILE=448 -DSWT_LANE_CAP=704"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  SWT_BPE_DEDUP=1 timeout -k 10 300 python bench.py --workload bpe_encode --corpus open --lean --steps 100 --warmup 10 > gpurun_out/r03ah.json 2> gpurun_out/r03ah.err || { tail -5 gpurun_out/r03ah.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03ah.json"))
print("[$v]:", d["value"], "MB/s", d["ms_per_step"], flush=True)
PY
done
