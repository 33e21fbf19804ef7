#!/bin/bash
# NOTE: this script measured a form that did not stay in the tree (see profiles/r03_experiments/); its build flag / environment knob exists only in the commit it ran against.
# dedup front end in two launches (SWT_DD_WARM = the share of tiles in the first one; 0 = one launch)
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "dedup or wp_ or config2 or config4 or headline_corpus_encode or smoke" > gpurun_out/r03z_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03z_pytest.log
if [ $rc -ne 0 ]; then head -80 gpurun_out/r03z_pytest.log; exit $rc; fi
one() {  # label, args
  local label=$1; shift
  timeout -k 10 400 python bench.py "$@" --lean > gpurun_out/r03z.json 2> gpurun_out/r03z.err || { tail -5 gpurun_out/r03z.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03z.json"))
print("$label:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
}
for w in 0 8 32 128; do
  SWT_DD_WARM=$w one "wp warm=$w" --workload wp_encode --steps 20 --warmup 5
done
for w in 0 32; do
  SWT_DD_WARM=$w SWT_BPE_DEDUP=2 one "lex warm=$w" --workload bpe_encode --corpus lex --steps 50 --warmup 10
  SWT_DD_WARM=$w one "mixed warm=$w" --workload mixed_encode --steps 10 --warmup 3
done
