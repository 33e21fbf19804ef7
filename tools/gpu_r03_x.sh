#!/bin/bash
# NOTE: this script measured a form that did not stay in the tree (see profiles/r03_experiments/); its build flag / environment knob exists only in the commit it ran against.
# bpe_lane_kernel: the pipelined round (scan while the lookups fly) against the plain order; dedup never / always per corpus
set -o pipefail
export TMPDIR=/tmp
one() {  # label, corpus
  timeout -k 10 300 python bench.py --workload bpe_encode --corpus $2 --lean --steps 100 --warmup 20 > gpurun_out/r03x.json 2> gpurun_out/r03x.err || { tail -5 gpurun_out/r03x.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03x.json"))
print("$1 $2:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
}
for v in "-DSWT_LANE_PIPE=1" "-DSWT_LANE_PIPE=0"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  SWT_BPE_DEDUP=1 one "[$v] never" open && SWT_BPE_DEDUP=1 one "[$v] never" lex && SWT_BPE_DEDUP=2 one "[$v] always" lex && SWT_BPE_DEDUP=0 one "[$v] auto" open
done
