# full GPU pass: all gpu tests, then the three bench workloads
export TMPDIR=/tmp
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 240 2>&1 | tail -4
for w in bpe_encode wp_encode bpe_train wp_train; do
  timeout -k 10 400 python bench.py --workload $w > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; echo "$w exit=$?"
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_$w.json"))
    print("$w", d["value"], d["unit"], "ms/step", d["ms_per_step"], "kernel_us", d["roofline"]["kernel_us"], "frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"])
except Exception as e:
    print("$w: no json", e); print(open("gpurun_out/bench_$w.err").read()[-1500:])
PY
done
