# dev loop for the BPE encode kernel: parity subset, bench line, ablation
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 240 -k "bpe" 2>&1 | tail -3
timeout -k 10 300 python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MB/s', d['value'], 'kernel_us', d['roofline']['kernel_us'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])"
timeout -k 10 300 python tools/gpu_ablate.py 2>&1 | grep "knob="
