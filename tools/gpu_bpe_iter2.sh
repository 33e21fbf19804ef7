export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 240 -k "bpe" 2>&1 | tail -3
timeout -k 10 300 python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'MB/s ms/step', d['ms_per_step'], 'kernel_us', d['roofline']['kernel_us'])"
cd /tmp; rm -rf /tmp/tp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 > /tmp/tp.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("/tmp/tp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-34s calls %5s avg_us %9.2f" % (r["Name"].split("(")[0][-34:], r["Calls"], float(r["AverageNs"])/1e3))
PY
