export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pm
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d /tmp/pm -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > /tmp/pm.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-30:]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "swt" not in k: continue
    print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
PY
