# rocprofv3 summaries for profiles/: kernel-trace stats per workload, then PMC passes (separate runs, no trace domains mixed in)
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01
rm -rf $O; mkdir -p $O
for w in bpe_encode wp_encode bpe_train; do
  extra=""; [ $w = bpe_encode ] && extra="--steps 50 --warmup 5"; [ $w = wp_encode ] && extra="--steps 5 --warmup 1"; [ $w = bpe_train ] && extra="--steps 1 --warmup 0"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python3 $R/bench.py --workload $w $extra > $O/$w.json 2> $O/$w.err; echo "$w trace exit=$?"
  f=$(find $O/$w -name "*kernel_stats.csv" | head -1); cp "$f" $O/${w}_kernel_stats.csv; cut -d, -f1-4 "$f" | cut -c1-110 | head -8
done
for w in bpe_encode wp_encode; do
for c in FETCH_SIZE WRITE_SIZE; do
  extra="--steps 5 --warmup 1"; [ $w = wp_encode ] && extra="--steps 3 --warmup 1"
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${w}_$c -- python3 $R/bench.py --workload $w $extra > $O/pmc_${w}_$c.json 2> $O/pmc_${w}_$c.err; echo "pmc $w $c exit=$?"
  f=$(find $O/pmc_${w}_$c -name "*counter_collection.csv" | head -1); echo $f
  python3 - "$f" $c $O $w <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "swt::" in r.get("Kernel_Name", "") and r.get("Counter_Name") == sys.argv[2]]
per = collections.defaultdict(list)
for r in rows:
    per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
calls = max([len(v) for k, v in per.items() if "wordref_kernel" in k] or [1])
tot = 0.0
out = "%s/%s_%s_per_kernel.csv" % (sys.argv[3], sys.argv[4], sys.argv[2])
with open(out, "w") as o:
    o.write("kernel,launches,mean_%s_KiB_per_launch\n" % sys.argv[2])
    for k, v in sorted(per.items()):
        o.write("%s,%d,%.1f\n" % (k, len(v), sum(v) / len(v)))
        tot += sum(v)
    o.write("ALL swt kernels per call (%d calls),,%.1f\n" % (calls, tot / calls))
print(open(out).read())
PY
  rm -rf $O/pmc_${w}_$c
done
done
rm -rf $O/bpe_encode $O/wp_encode $O/bpe_train
ls -la $O
