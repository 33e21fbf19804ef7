#!/bin/bash
# bpe_lane_kernel with lane refill: tile / chunk sweep again (rebuilds on the box), then the kernel trace of the default
set -o pipefail
export TMPDIR=/tmp
for v in "-DSWT_LANE_TILE=320 -DSWT_LANE_CAP=448" "-DSWT_LANE_TILE=384 -DSWT_LANE_CAP=512" "-DSWT_LANE_TILE=512 -DSWT_LANE_CAP=640" "-DSWT_LANE_TILE=640 -DSWT_LANE_CAP=768" "-DSWT_LANE_TILE=768 -DSWT_LANE_CAP=1024" "-DSWT_LANE_TILE=1024 -DSWT_LANE_CAP=1280" "-DSWT_LANE_TILE=1536 -DSWT_LANE_CAP=2048"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  timeout -k 10 300 python bench.py --workload bpe_encode --corpus open --lean --steps 50 --warmup 5 > gpurun_out/r03s_sweep.json 2> gpurun_out/r03s_sweep.err || { tail -5 gpurun_out/r03s_sweep.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03s_sweep.json"))
print("$v:", d["value"], "MB/s", d["ms_per_step"], flush=True)
PY
done
unset SWT_EXTRA_FLAGS
python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
cd /tmp && rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --workload bpe_encode --corpus open --lean --steps 30 --warmup 20 > /tmp/kt.log 2>&1; f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r03s_open_kernel_stats.csv; cut -d, -f1-4 $f | cut -c1-60,200- | head -12
