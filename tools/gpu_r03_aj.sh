#!/bin/bash
# NOTE: this script measured a form that did not stay in the tree (see profiles/r03_experiments/); its build flag / environment knob exists only in the commit it ran against.
# bpe_lane_kernel tail: a lane's share of the slots is four when that covers the word
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "word_lane or twin or bpe_edge or bpe_fuzz or random_tables or headline_corpus_encode" 2>&1 | tail -1
for c in open lex open; do
  SWT_BPE_DEDUP=1 timeout -k 10 300 python bench.py --workload bpe_encode --corpus $c --lean --steps 100 --warmup 10 > gpurun_out/r03aj.json 2> gpurun_out/r03aj.err || { tail -5 gpurun_out/r03aj.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03aj.json"))
print("$c:", d["value"], "MB/s", d["ms_per_step"], flush=True)
PY
done
