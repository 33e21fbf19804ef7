# quick loop: encode parity tests (BPE + WP), then per-call time of both encoders
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 240 -k "bpe or wp_" 2>&1 | tail -3
for w in bpe_encode wp_encode; do
  timeout -k 10 300 python bench.py --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['value'], 'MB/s ms/step', d['ms_per_step'], 'call_us', r['kernel_us'], 'dominant', r.get('dominant_kernel'))"
done
