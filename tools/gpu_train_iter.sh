export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 240 -k "train" 2>&1 | tail -3
timeout -k 10 300 python tools/gpu_train_prof.py 2>&1 | grep -E "run\(|stepwise|create"
timeout -k 10 300 python bench.py --workload bpe_train 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], 'ms/step', d['ms_per_step'], d['roofline']['kernel_us'])"
