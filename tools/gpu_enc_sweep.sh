# encode tile / chunk sweep (SWT_EXTRA_FLAGS rebuilds on the box)
export TMPDIR=/tmp
for v in "${@}"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  for c in open lex; do
    timeout -k 10 300 python bench.py --workload bpe_encode --corpus $c --lean --steps 50 --warmup 5 > gpurun_out/sweep_$c.json 2> gpurun_out/sweep.err || { tail -5 gpurun_out/sweep.err; exit 1; }
    python - <<PY
import json
d=json.load(open("gpurun_out/sweep_$c.json"))
print("$v $c:", d["value"], "MB/s", d["ms_per_step"], flush=True)
PY
  done
done
