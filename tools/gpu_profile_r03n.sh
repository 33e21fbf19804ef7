# round 3, final tree (word-lane FastBPE kernel, cuckoo rank table, cached first look in the dedup table): rocprofv3 summaries behind
# the default bench line and the per-workload lines, the PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, --pmc only), and the SQ
# counters of the FastBPE direct path.  Everything lands in gpurun_out/prof_r03n/; what is judged is copied to profiles/r03n_*.
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03n
rm -rf $O; mkdir -p $O
run_trace() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name trace exit=$?"
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/${name}_kernel_stats.csv; cut -d, -f1-4 "$f" | cut -c1-50,230- | head -5
  rm -rf $O/$name
}
run_trace headline --steps 20 --warmup 5
run_trace bpe_encode_open --workload bpe_encode --corpus open --steps 40 --warmup 70 --lean
run_trace wp_encode --workload wp_encode --steps 10 --warmup 2
run_trace mixed_encode --workload mixed_encode --steps 5 --warmup 2
run_pmc() {  # name, counter, bench args...
  local name=$1 c=$2; shift; shift
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${name}_$c -- python3 $R/bench.py "$@" > $O/pmc_${name}_$c.json 2> $O/pmc_${name}_$c.err; echo "pmc $name $c exit=$?"
  f=$(find $O/pmc_${name}_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c $O $name <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "swt::" in r.get("Kernel_Name", "") and r.get("Counter_Name") == sys.argv[2]]
per = collections.defaultdict(list)
for r in rows:
    per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
out = "%s/%s_%s_per_kernel.csv" % (sys.argv[3], sys.argv[4], sys.argv[2])
with open(out, "w") as o:
    o.write("kernel,launches,mean_%s_KiB_per_launch,total_KiB\n" % sys.argv[2])
    tot = 0.0
    for k, v in sorted(per.items()):
        o.write("%s,%d,%.1f,%.1f\n" % (k, len(v), sum(v) / len(v), sum(v)))  # (kernel names hold commas -- template arguments: read from the right)
        tot += sum(v)
    o.write("ALL swt kernels,,,%.1f\n" % tot)
print(open(out).read()[:1500])
PY
  rm -rf $O/pmc_${name}_$c
}
for c in FETCH_SIZE WRITE_SIZE; do
  run_pmc wp_encode $c --workload wp_encode --steps 4 --warmup 1 --lean
  run_pmc mixed_encode $c --workload mixed_encode --steps 4 --warmup 1 --lean
  SWT_BPE_DEDUP=1 run_pmc bpe_encode_open $c --workload bpe_encode --corpus open --steps 8 --warmup 2 --lean
  run_pmc bpe_encode_lex $c --workload bpe_encode --corpus lex --steps 8 --warmup 2 --lean
done
SQ=$O/sq_counters.txt
: > $SQ
run_sq() {  # label, counters, bench args...
  local label=$1 ctr=$2; shift; shift
  rm -rf /tmp/pm
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pm -- python3 $R/bench.py "$@" > /tmp/pm.log 2>&1 || { tail -5 /tmp/pm.log; return 1; }
  python3 - "$label" >> $SQ <<'PY'
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
SWT_BPE_DEDUP=1 run_sq bpe_open "$A" --workload bpe_encode --corpus open --steps 8 --warmup 2 --lean
SWT_BPE_DEDUP=1 run_sq bpe_open "$B" --workload bpe_encode --corpus open --steps 8 --warmup 2 --lean
run_sq wp_encode "$A" --workload wp_encode --steps 4 --warmup 1 --lean
run_sq wp_encode "$B" --workload wp_encode --steps 4 --warmup 1 --lean
cut -c1-380 $SQ
ls $O
cd $R
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo default_bench_exit=$?
