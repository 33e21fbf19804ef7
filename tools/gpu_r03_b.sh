#!/bin/bash
# round 3, second GPU call: WordPiece training after the index tie-break, the sharded runner's timings, the dedup tile sweep
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "wp_train or sharded or separators or smoke or collision" > gpurun_out/r03b_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03b_pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --workload wp_train > gpurun_out/r03b_wp_train.json 2> gpurun_out/r03b_wp_train.err; echo "wp_train rc $?"
python -c "import json; d=json.load(open('gpurun_out/r03b_wp_train.json')); print(d['value'], d['roofline']['kernel_us'], d['roofline']['bytes_model'])"
timeout -k 10 500 python tools/gpu_sharded_bench.py > gpurun_out/r03b_sharded.jsonl 2> gpurun_out/r03b_sharded.err; rc=$?; echo "sharded rc $rc"; cat gpurun_out/r03b_sharded.jsonl; tail -3 gpurun_out/r03b_sharded.err
if [ $rc -ge 124 ]; then exit $rc; fi
bash tools/gpu_dd_sweep.sh "" "-DSWT_DTILE=512 -DSWT_DCAP=1024" "-DSWT_DTILE=768 -DSWT_DCAP=1280" "-DSWT_DTILE=1536 -DSWT_DCAP=2048" 2>&1 | tee gpurun_out/r03b_ddsweep.txt
