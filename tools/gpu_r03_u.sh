#!/bin/bash
# NOTE: this script measured a form that did not stay in the tree (see profiles/r03_experiments/); its build flag / environment knob exists only in the commit it ran against.
# two-launch form (self-planning tiles + gather_blocks): diagnosis, BPE parity tests, bench against the four-launch form
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python tools/gpu_dbg_bpe.py 2>&1 | tail -8 || exit 1
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "bpe_ or smoke or cli or dedup or headline_corpus_encode or single_launch or random_tables or lowercase or joined or config4_mixed" > gpurun_out/r03u_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03u_pytest.log
if [ $rc -ne 0 ]; then head -80 gpurun_out/r03u_pytest.log; exit $rc; fi
one() {  # label, corpus
  timeout -k 10 300 python bench.py --workload bpe_encode --corpus $2 --lean --steps 100 --warmup 20 > gpurun_out/r03u_$1.json 2> gpurun_out/r03u_$1.err || { tail -5 gpurun_out/r03u_$1.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03u_$1.json"))
print("$1 $2:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
}
one piped open && one piped_lex lex
