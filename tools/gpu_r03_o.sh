#!/bin/bash
# bpe_lane_kernel: SQ counters on S85k-open, then a tile / chunk sweep (rebuilds on the box)
set -o pipefail
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03o_sq.txt
: > $OUT
cd /tmp
run() {  # label, counters, bench args...
  local label=$1 ctr=$2; shift; shift
  rm -rf /tmp/pm
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pm -- python3 $ROOT/bench.py "$@" > /tmp/pm.log 2>&1 || { tail -5 /tmp/pm.log; return 1; }
  python3 - "$label" >> $OUT <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if "swt::" not in k: continue
    n = max(len(v) for v in d.values())
    if n < 4: continue
    print(sys.argv[1], "|", k[-44:], "| launches", n, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
}
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES"
run open "$A" --workload bpe_encode --corpus open --steps 4 --warmup 1 --lean && \
run open "$B" --workload bpe_encode --corpus open --steps 4 --warmup 1 --lean && \
run open "$C" --workload bpe_encode --corpus open --steps 4 --warmup 1 --lean
cat $OUT | cut -c1-400
cd $ROOT
for v in "-DSWT_LANE_TILE=192 -DSWT_LANE_CAP=256" "-DSWT_LANE_TILE=256 -DSWT_LANE_CAP=384" "-DSWT_LANE_TILE=320 -DSWT_LANE_CAP=448" "-DSWT_LANE_TILE=384 -DSWT_LANE_CAP=512" "-DSWT_LANE_TILE=512 -DSWT_LANE_CAP=640" "-DSWT_LANE_TILE=768 -DSWT_LANE_CAP=1024"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  timeout -k 10 300 python bench.py --workload bpe_encode --corpus open --lean --steps 50 --warmup 5 > gpurun_out/r03o_sweep.json 2> gpurun_out/r03o_sweep.err || { tail -5 gpurun_out/r03o_sweep.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03o_sweep.json"))
print("$v:", d["value"], "MB/s", d["ms_per_step"], flush=True)
PY
done
