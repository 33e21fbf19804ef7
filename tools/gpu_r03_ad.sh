#!/bin/bash
# wp_refs_kernel: records kept in registers between the count loop and the copy, first three tokens of a run loaded together
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "wp_ or config2 or config4 or smoke" > gpurun_out/r03ad_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03ad_pytest.log
if [ $rc -ne 0 ]; then head -80 gpurun_out/r03ad_pytest.log; exit $rc; fi
cd /tmp && rm -rf /tmp/kt && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --workload wp_encode --steps 20 --warmup 5 --lean > $GRAFT_REPO_ROOT/gpurun_out/r03ad.json 2> /tmp/kt.err
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
python3 - $f $GRAFT_REPO_ROOT/gpurun_out/r03ad.json <<'PY'
import csv, sys, json
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(r["Name"][:56].ljust(56), r["Calls"], r["AverageNs"], r["Percentage"])
d=json.load(open(sys.argv[2])); print(d["value"], d["ms_per_step"])
PY
