# SQ counters per kernel, round 3: the FastWP call (wordref / refs) and WordPiece training (wp_step / apply); two passes per
# workload (instruction mix, then wait / active cycles); --pmc only, never with a trace domain
export TMPDIR=/tmp
cd /tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sq_r03.txt
: > $OUT
run() {  # label, counters, bench args...
  local label=$1 ctr=$2; shift; shift
  rm -rf /tmp/pm
  timeout -k 10 400 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > /tmp/pm.log 2>&1 || { tail -5 /tmp/pm.log; return 1; }
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
run wp_encode "$A" --workload wp_encode --steps 4 --warmup 1 --lean && \
run wp_encode "$B" --workload wp_encode --steps 4 --warmup 1 --lean && \
run wp_train "$A" --workload wp_train --steps 1 && \
run wp_train "$B" --workload wp_train --steps 1
cat $OUT | cut -c1-420
