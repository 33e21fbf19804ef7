# quick loop: encode parity tests (BPE + WP), then per-call time of both encoders
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 240 -k "bpe or wp_" 2>&1 | tail -3
for w in bpe_encode wp_encode; do
  timeout -k 10 300 python bench.py --workload $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['value'], 'MB/s ms/step', d['ms_per_step'], 'call_us', r['kernel_us'], 'dominant', r.get('dominant_kernel'))"
done
cd /tmp; rm -rf /tmp/tp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -- python3 $GRAFT_REPO_ROOT/bench.py --workload wp_encode --steps 5 --warmup 1 > /tmp/tp.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("/tmp/tp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-40s calls %5s avg_us %9.2f" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"])/1e3))
PY
