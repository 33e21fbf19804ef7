# round 2: training tests, at-spec config tests, training benches
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q --timeout 400 -k "train or sharded or config3 or collision" 2>&1 | tail -15
for w in "bpe_train" "bpe_train --corpus lex" "wp_train" "bpe_train_1g"; do
  n=$(echo $w | tr ' -' '__')
  timeout -k 10 500 python bench.py --workload $w > gpurun_out/r02_$n.json 2> gpurun_out/r02_$n.err; echo "$w exit=$?"
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02_$n.json"))
    r=d["roofline"]
    print("$w", d["value"], d["unit"], "ms/step", d["ms_per_step"], "merge_us", r["kernel_us"], "frac", r["frac"], "bytes/merge", r["algorithmic_bytes_per_launch"], r["bytes_model"], "merges/step", r["merges_per_step"], "rescan eff GB/s", r["rescan_formulation"]["effective_gbs"], "cpu", d["cpu_baseline"]["value"])
except Exception as e:
    print("$w: no json", e); print(open("gpurun_out/r02_$n.err").read()[-1500:])
PY
done
