#!/bin/bash
# NOTE: this script measured a form that did not stay in the tree (see profiles/r03_experiments/); its build flag / environment knob exists only in the commit it ran against.
# bpe_lane_kernel tail: four lanes a word, then eight (SWT_TAIL_WIDE) -- rebuilds on the box
set -o pipefail
export TMPDIR=/tmp
for v in "-DSWT_TAIL_WIDE=1" "-DSWT_TAIL_WIDE=0"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  timeout -k 10 200 python -m pytest tests -m gpu -q -x -k "word_lane or twin or bpe_edge or bpe_fuzz" 2>&1 | tail -1
  for c in open lex; do
  SWT_BPE_DEDUP=1 timeout -k 10 300 python bench.py --workload bpe_encode --corpus $c --lean --steps 100 --warmup 10 > gpurun_out/r03ai.json 2> gpurun_out/r03ai.err || { tail -5 gpurun_out/r03ai.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03ai.json"))
print("[$v] $c:", d["value"], "MB/s", d["ms_per_step"], flush=True)
PY
  done
done
