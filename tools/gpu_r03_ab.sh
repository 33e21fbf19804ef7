#!/bin/bash
# NOTE: this script measured a form that did not stay in the tree (see profiles/r03_experiments/); its build flag / environment knob exists only in the commit it ran against.
# dedup word table with the word's first 14 bytes in the slot (SWT_DD_NO_PREFIX=1: every compare goes to the text, as before)
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "dedup or wp_ or config2 or config4 or headline_corpus_encode or smoke or word_lane" > gpurun_out/r03ab_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03ab_pytest.log
if [ $rc -ne 0 ]; then head -80 gpurun_out/r03ab_pytest.log; exit $rc; fi
one() {  # label, args
  local label=$1; shift
  timeout -k 10 400 python bench.py "$@" --lean > gpurun_out/r03ab.json 2> gpurun_out/r03ab.err || { tail -5 gpurun_out/r03ab.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03ab.json"))
print("$label:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
}
one "wp prefix" --workload wp_encode --steps 20 --warmup 5 && SWT_DD_NO_PREFIX=1 one "wp text-compare" --workload wp_encode --steps 20 --warmup 5
SWT_BPE_DEDUP=2 one "lex prefix (dedup forced)" --workload bpe_encode --corpus lex --steps 50 --warmup 10 && SWT_DD_NO_PREFIX=1 SWT_BPE_DEDUP=2 one "lex text-compare (dedup forced)" --workload bpe_encode --corpus lex --steps 50 --warmup 10
one "mixed prefix" --workload mixed_encode --steps 10 --warmup 3 && SWT_DD_NO_PREFIX=1 one "mixed text-compare" --workload mixed_encode --steps 10 --warmup 3
