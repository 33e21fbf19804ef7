#!/bin/bash
# tile_scan_kernel without device-scope fences (the totals in device-scope stores / loads); wp_step_kernel likewise
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "bpe_ or wp_ or dedup or smoke or cli or headline_corpus_encode or config2 or config4 or lowercase or joined or word_lane or pack_and" > gpurun_out/r03ao_pytest.log 2>&1
rc=$?; tail -2 gpurun_out/r03ao_pytest.log
if [ $rc -ne 0 ]; then head -60 gpurun_out/r03ao_pytest.log; exit $rc; fi
one() {  # label, args
  local label=$1; shift
  timeout -k 10 400 python bench.py "$@" --lean > gpurun_out/r03ao.json 2> gpurun_out/r03ao.err || { tail -5 gpurun_out/r03ao.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03ao.json"))
print("$label:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
}
SWT_BPE_DEDUP=1 one "open direct" --workload bpe_encode --corpus open --steps 100 --warmup 20
one "wp" --workload wp_encode --steps 20 --warmup 5
one "mixed" --workload mixed_encode --steps 10 --warmup 3
SWT_BPE_DEDUP=1 one "open direct (again)" --workload bpe_encode --corpus open --steps 100 --warmup 20
