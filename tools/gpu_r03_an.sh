#!/bin/bash
# wp_step_kernel: workgroup-scope fences around the ticket instead of __threadfence()
set -o pipefail
export TMPDIR=/tmp
for v in "" "-DSWT_WP_THREADFENCE"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  timeout -k 10 400 python -m pytest tests -m gpu -q -x -k "wp_train or collision_replay" 2>&1 | tail -1
  timeout -k 10 300 python bench.py --workload wp_train > gpurun_out/r03an.json 2> gpurun_out/r03an.err || { tail -5 gpurun_out/r03an.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03an.json"))
print("[$v]", d.get("value"), d.get("unit"), {k: d[k] for k in d if "us_per" in k or "merge" in k}, flush=True)
PY
done
