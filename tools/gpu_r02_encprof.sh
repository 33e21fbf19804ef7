export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_enc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_enc -- python3 bench.py --workload ${1:-bpe_encode} --steps 20 --warmup 3 > gpurun_out/prof_enc.json 2> gpurun_out/prof_enc.err; echo prof_exit=$?
f=$(find gpurun_out/prof_enc -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-70s calls %6s avg_us %9.2f min %8.2f max %9.2f pct %s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, r["Percentage"]))
PY
