#!/bin/bash
# bpe_lane_kernel with the four-lane tail: tile / chunk sweep again (rebuilds on the box)
set -o pipefail
export TMPDIR=/tmp
for v in "-DSWT_LANE_TILE=384 -DSWT_LANE_CAP=512" "-DSWT_LANE_TILE=512 -DSWT_LANE_CAP=640" "-DSWT_LANE_TILE=576 -DSWT_LANE_CAP=704" "-DSWT_LANE_TILE=640 -DSWT_LANE_CAP=768" "-DSWT_LANE_TILE=768 -DSWT_LANE_CAP=1024" "-DSWT_LANE_TILE=1024 -DSWT_LANE_CAP=1280"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  for c in open lex; do
  SWT_BPE_DEDUP=1 timeout -k 10 300 python bench.py --workload bpe_encode --corpus $c --lean --steps 100 --warmup 10 > gpurun_out/r03ak.json 2> gpurun_out/r03ak.err || { tail -5 gpurun_out/r03ak.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03ak.json"))
print("[$v] $c:", d["value"], "MB/s", d["ms_per_step"], flush=True)
PY
  done
done
