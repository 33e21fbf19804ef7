#!/bin/bash
# dedup front end: cached first look at the word table against the coherent-only form (SWT_DD_COHERENT=1); WP + BPE-lex + mixed
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "dedup or wp_ or config2 or config4 or headline_corpus_encode or smoke" > gpurun_out/r03y_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03y_pytest.log
if [ $rc -ne 0 ]; then head -80 gpurun_out/r03y_pytest.log; exit $rc; fi
one() {  # label, args
  local label=$1; shift
  timeout -k 10 400 python bench.py "$@" --lean > gpurun_out/r03y.json 2> gpurun_out/r03y.err || { tail -5 gpurun_out/r03y.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03y.json"))
print("$label:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
}
one "wp cached" --workload wp_encode --steps 20 --warmup 5 && SWT_DD_COHERENT=1 one "wp coherent" --workload wp_encode --steps 20 --warmup 5
SWT_BPE_DEDUP=2 one "lex cached (dedup forced)" --workload bpe_encode --corpus lex --steps 50 --warmup 10 && SWT_DD_COHERENT=1 SWT_BPE_DEDUP=2 one "lex coherent (dedup forced)" --workload bpe_encode --corpus lex --steps 50 --warmup 10
one "mixed cached" --workload mixed_encode --steps 10 --warmup 3 && SWT_DD_COHERENT=1 one "mixed coherent" --workload mixed_encode --steps 10 --warmup 3
