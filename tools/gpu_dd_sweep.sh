# dedup tile / chunk sweep for the FastWP call (SWT_EXTRA_FLAGS rebuilds on the box): usage  bash tools/gpu_dd_sweep.sh "" "-DSWT_DTILE=512 -DSWT_DCAP=1024" ...
export TMPDIR=/tmp
for v in "${@}"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  timeout -k 10 300 python bench.py --workload wp_encode --lean --steps 10 --warmup 2 > gpurun_out/ddsweep.json 2> gpurun_out/ddsweep.err || { tail -5 gpurun_out/ddsweep.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ddsweep.json"))
print("[$v] wp_encode:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
  timeout -k 10 300 python bench.py --workload bpe_encode --corpus lex --lean --steps 50 --warmup 5 > gpurun_out/ddsweep.json 2> gpurun_out/ddsweep.err || { tail -5 gpurun_out/ddsweep.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ddsweep.json"))
print("[$v] bpe_encode lex:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
done
