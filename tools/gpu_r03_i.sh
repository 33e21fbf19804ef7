#!/bin/bash
# soak again: the seed that met the table bound, then a new one; then the training lines (the bound sizes the pair table)
set -o pipefail
mkdir -p gpurun_out
SWT_SOAK_SEED=12787955 SWT_SOAK_SECONDS=200 timeout -k 10 400 python tools/gpu_soak.py > gpurun_out/r03i_soak_a.txt 2>&1; rc=$?
tail -2 gpurun_out/r03i_soak_a.txt
if [ $rc -ne 0 ]; then exit $rc; fi
SWT_SOAK_SECONDS=200 timeout -k 10 400 python tools/gpu_soak.py > gpurun_out/r03i_soak_b.txt 2>&1; rc=$?
tail -2 gpurun_out/r03i_soak_b.txt
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --workload bpe_train --steps 2 > gpurun_out/r03i_bpe_train.json 2> gpurun_out/r03i_bpe_train.err; echo "rc $?"
python -c "import json; d=json.load(open('gpurun_out/r03i_bpe_train.json')); print('bpe_train', d['value'], d['us_per_merge_device'], d['ms_per_step'])"
timeout -k 10 400 python bench.py --workload bpe_train_1g > gpurun_out/r03i_bpe_train_1g.json 2> gpurun_out/r03i_bpe_train_1g.err; echo "rc $?"
python -c "import json; d=json.load(open('gpurun_out/r03i_bpe_train_1g.json')); print('bpe_train_1g', d['value'], d['roofline']['kernel_us'], d['ms_per_step'])"
timeout -k 10 400 python bench.py --workload mixed_encode > gpurun_out/r03i_mixed.json 2> gpurun_out/r03i_mixed.err; echo "rc $?"
python -c "import json; d=json.load(open('gpurun_out/r03i_mixed.json')); print('mixed (two streams)', d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline'].get('sum_of_call_spans_us'))"
