#!/bin/bash
# dedup word table: size sweep now that the first look goes through the caches (SWT_DD_BITS = log2 slots; default: one slot per 32 bytes)
set -o pipefail
export TMPDIR=/tmp
one() {  # label, args
  local label=$1; shift
  timeout -k 10 400 python bench.py "$@" --lean > gpurun_out/r03ac.json 2> gpurun_out/r03ac.err || { tail -5 gpurun_out/r03ac.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03ac.json"))
print("$label:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
}
for b in 0 17 18 19 20 21; do
  SWT_DD_BITS=$b one "wp bits=$b" --workload wp_encode --steps 20 --warmup 5
done
for b in 0 17 18 19; do
  SWT_DD_BITS=$b one "mixed bits=$b" --workload mixed_encode --steps 10 --warmup 3
done
