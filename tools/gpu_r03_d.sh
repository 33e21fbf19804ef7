#!/bin/bash
# WordPiece training: where does a merge's time go?  kernel trace (durations), host enqueue time per trip
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
SWT_TRAIN_DEBUG=1 timeout -k 10 200 python bench.py --workload wp_train > gpurun_out/r03d_wp_train.json 2> gpurun_out/r03d_wp_train.err; echo "rc $?"
grep "^trip" gpurun_out/r03d_wp_train.err | tail -8
python -c "import json; d=json.load(open('gpurun_out/r03d_wp_train.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'])"
SWT_TRAIN_DEBUG=1 timeout -k 10 200 python bench.py --workload bpe_train --steps 1 > gpurun_out/r03d_bpe_train.json 2> gpurun_out/r03d_bpe_train.err; echo "rc $?"
grep "^trip" gpurun_out/r03d_bpe_train.err | tail -5
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wpt -- python3 $R/bench.py --workload wp_train --steps 1 > /tmp/wpt.json 2> /tmp/wpt.err; echo "trace rc $?"
f=$(find /tmp/wpt -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r03d_wp_train_kernel_stats.csv; cut -d, -f1-6 "$f" | cut -c1-160 | head -12
