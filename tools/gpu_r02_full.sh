# round 2: the whole GPU suite, smoke, the default bench line
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 2>&1 | tail -6
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_headline.json 2> gpurun_out/r02_headline.err; echo bench_exit=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02_headline.json"))
print("value", d["value"], d["unit"], d["config"]["workload"][:60])
print("roofline", d["roofline"]["frac"], d["roofline"]["kernel_us"], "other", d["other_corpus"]["value"], d["other_corpus"]["roofline"]["kernel_us"])
print("train", {k: d["train"][k] for k in ("s_per_1k_merges","us_per_merge_device","train_wall_s","n_merges")}, d["train"]["roofline"]["frac"], d["train"]["cpu_baseline"]["value"])
print("detail", d["encode_detail"])
PY
