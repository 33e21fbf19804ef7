#!/bin/bash
# bpe_lane_kernel: the split's pair lookups taken up one block later (software-pipelined)
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "bpe_ or smoke or cli or dedup or headline_corpus_encode or single_launch or random_tables or lowercase or joined or config4_mixed or word_lane" > gpurun_out/r03ag_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03ag_pytest.log
if [ $rc -ne 0 ]; then head -80 gpurun_out/r03ag_pytest.log; exit $rc; fi
one() {  # label, args
  local label=$1; shift
  timeout -k 10 400 python bench.py "$@" --lean > gpurun_out/r03ag.json 2> gpurun_out/r03ag.err || { tail -5 gpurun_out/r03ag.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03ag.json"))
print("$label:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
}
SWT_BPE_DEDUP=1 one "open direct" --workload bpe_encode --corpus open --steps 100 --warmup 20
SWT_BPE_DEDUP=1 one "lex direct" --workload bpe_encode --corpus lex --steps 100 --warmup 20
SWT_BPE_DEDUP=1 one "open direct (again)" --workload bpe_encode --corpus open --steps 100 --warmup 20
