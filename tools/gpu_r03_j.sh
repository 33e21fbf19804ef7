#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "sharded or multi" > gpurun_out/r03j_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03j_pytest.log
if [ $rc -ne 0 ]; then head -60 gpurun_out/r03j_pytest.log; exit $rc; fi
/usr/bin/time -v timeout -k 10 600 python bench.py > gpurun_out/r03j_bench.json 2> gpurun_out/r03j_bench.err; echo "bench rc $?"
grep -E "Elapsed|Maximum resident" gpurun_out/r03j_bench.err
python -c "import json; d=json.load(open('gpurun_out/r03j_bench.json')); print(d['value'], d['wp_encode']['value'], d['mixed_encode']['value'], d['train']['s_per_1k_merges'])"
