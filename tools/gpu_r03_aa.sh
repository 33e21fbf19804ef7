#!/bin/bash
# FastWP call after the cached first look: kernel trace and the FETCH_SIZE / WRITE_SIZE passes
export TMPDIR=/tmp
cd /tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03aa
rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload wp_encode --steps 10 --warmup 3 --lean > $O/kt.json 2> $O/kt.err
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); cp "$f" $O/wp_encode_kernel_stats.csv; rm -rf $O/kt
python3 - $O/wp_encode_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:56].ljust(56), r["Calls"], r["AverageNs"], r["Percentage"])
PY
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --workload wp_encode --steps 4 --warmup 1 --lean > $O/pmc_$c.json 2> $O/pmc_$c.err
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c <<'PY'
import csv, sys, collections
per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "swt::" in r.get("Kernel_Name", "") and r.get("Counter_Name") == sys.argv[2]:
        per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(per.items()):
    print(sys.argv[2], k[-40:], len(v), round(sum(v) / len(v) / 1024, 1), "MiB per launch")
PY
  rm -rf $O/pmc_$c
done
