#!/bin/bash
# NOTE: this script measured a form that did not stay in the tree (see profiles/r03_experiments/); its build flag / environment knob exists only in the commit it ran against.
# bpe_lane_kernel (refill form): SQ counters, then the phase ablation
set -o pipefail
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r03t_sq.txt
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
    if n < 10: continue
    print(sys.argv[1], "|", k[-44:], "| launches", n, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
}
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
run open "$A" --workload bpe_encode --corpus open --steps 12 --warmup 20 --lean && \
run open "$B" --workload bpe_encode --corpus open --steps 12 --warmup 20 --lean
cat $OUT | cut -c1-400
cd $ROOT
bash tools/gpu_r03_p.sh
