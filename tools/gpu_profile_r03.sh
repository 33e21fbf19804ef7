# round 3: rocprofv3 summaries behind the default bench line (headline: FastBPE encode + wp_encode + mixed_encode + train blocks),
# the per-workload lines, and the PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, --pmc only) for the three encode workloads
# and training.  Everything lands in gpurun_out/prof_r03/; the summaries that are judged are copied to profiles/r03_*.
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
run_trace() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name trace exit=$?"
  f=$(find $O/$name -name "*kernel_stats.csv" | head -1); cp "$f" $O/${name}_kernel_stats.csv; cut -d, -f1-4 "$f" | cut -c1-120 | head -5
  rm -rf $O/$name
}
run_trace headline --steps 20 --warmup 5
run_trace wp_encode --workload wp_encode --steps 10 --warmup 2
run_trace mixed_encode --workload mixed_encode --steps 5 --warmup 2
run_trace wp_train --workload wp_train --steps 1
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
print(open(out).read()[:1800])
PY
  rm -rf $O/pmc_${name}_$c
}
for c in FETCH_SIZE WRITE_SIZE; do
  run_pmc wp_encode $c --workload wp_encode --steps 4 --warmup 1 --lean
  run_pmc mixed_encode $c --workload mixed_encode --steps 4 --warmup 1 --lean
  run_pmc bpe_encode_open $c --workload bpe_encode --corpus open --steps 8 --warmup 2 --lean
done
ls $O
cd $R
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo default_bench_exit=$?
timeout -k 10 300 python bench.py --workload wp_train > $O/bench_wp_train.json 2> $O/bench_wp_train.err; echo wp_train_exit=$?
timeout -k 10 400 python bench.py --workload bpe_train_1g > $O/bench_bpe_train_1g.json 2> $O/bench_bpe_train_1g.err; echo bpe_train_1g_exit=$?
SWT_BENCH_FORCE_SHARDED=1 timeout -k 10 300 python bench.py --workload bpe_train --steps 1 > $O/bench_forced_sharded_fast.json 2> $O/bench_forced_sharded_fast.err; echo forced_sharded_fast_exit=$?
SWT_BENCH_FORCE_SHARDED=1 SWT_DIST_GENERIC=1 timeout -k 10 300 python bench.py --workload bpe_train --steps 1 > $O/bench_forced_sharded_generic.json 2> $O/bench_forced_sharded_generic.err; echo forced_sharded_generic_exit=$?
