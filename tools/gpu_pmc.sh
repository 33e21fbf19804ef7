# SQ counters per kernel of the default bench (two passes: instruction mix, then wait/active cycles)
export TMPDIR=/tmp
cd /tmp
run() {
  rm -rf /tmp/pm
  timeout -k 10 300 rocprofv3 --pmc $1 --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > /tmp/pm.log 2>&1 || { tail -5 /tmp/pm.log; return 1; }
  python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-34:]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "swt" not in k: continue
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
}
run "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" && \
run "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU"
