#!/bin/bash
# back half of the dedup: records in registers (write launch only), first tokens of a run loaded together (FastWP and FastBPE)
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "wp_ or dedup or config2 or config4 or smoke or word_lane" > gpurun_out/r03ae_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03ae_pytest.log
if [ $rc -ne 0 ]; then head -80 gpurun_out/r03ae_pytest.log; exit $rc; fi
for w in wp_encode mixed_encode; do
cd /tmp && rm -rf /tmp/kt && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 10 --warmup 5 --lean > $GRAFT_REPO_ROOT/gpurun_out/r03ae_$w.json 2> /tmp/kt.err
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
python3 - $f $GRAFT_REPO_ROOT/gpurun_out/r03ae_$w.json <<'PY'
import csv, sys, json
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(r["Name"][:56].ljust(56), r["Calls"], r["AverageNs"], r["Percentage"])
d=json.load(open(sys.argv[2])); print(d["value"], d["ms_per_step"])
PY
done
cd $GRAFT_REPO_ROOT && SWT_BPE_DEDUP=2 timeout -k 10 300 python bench.py --workload bpe_encode --corpus lex --lean --steps 50 --warmup 10 > gpurun_out/r03ae_lex.json && python -c "
import json; d=json.load(open('gpurun_out/r03ae_lex.json')); print('lex dedup forced', d['value'], d['ms_per_step'])"
