#!/bin/bash
# full -m gpu suite on the current tree, then the randomized differential soak (training, both encoders, sharded runner, WordPiece training)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03h_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03h_pytest.log
if [ $rc -ne 0 ]; then head -60 gpurun_out/r03h_pytest.log; exit $rc; fi
SWT_SOAK_SECONDS=240 timeout -k 10 400 python tools/gpu_soak.py > gpurun_out/r03h_soak.txt 2>&1; rc=$?
tail -4 gpurun_out/r03h_soak.txt
exit $rc
