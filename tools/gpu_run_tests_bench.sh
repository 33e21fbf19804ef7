# GPU-box driver used during development: bench line, rocprofv3 kernel stats, gpu tests.
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo bench_exit=$?
cat gpurun_out/bench.json
rm -rf gpurun_out/prof && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 50 --warmup 5 > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; echo prof_exit=$?
cut -d, -f1-4 $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) | cut -c1-150
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -v --timeout 240 2>&1 | tee gpurun_out/pytest_gpu.log | grep -E "PASSED|FAILED|ERROR|passed|failed|Timeout" ; echo pytest_exit=${PIPESTATUS[0]}
