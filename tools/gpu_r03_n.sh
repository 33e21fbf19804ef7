#!/bin/bash
# the word-lane kernel (bpe_lane_kernel): BPE parity tests first, then the S85k-open / S85k-lex bench with both kernels
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "bpe_ or smoke or cli or dedup or headline_corpus_encode or single_launch or random_tables or lowercase or joined or config4_mixed" > gpurun_out/r03n_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03n_pytest.log
if [ $rc -ne 0 ]; then head -80 gpurun_out/r03n_pytest.log; exit $rc; fi
for kern in lane bytes; do
  export SWT_BPE_KERNEL=$kern
  for c in open lex; do
    timeout -k 10 300 python bench.py --workload bpe_encode --corpus $c > gpurun_out/r03n_bpe_${c}_$kern.json 2> gpurun_out/r03n_bpe_${c}_$kern.err; echo "$kern $c rc $?"
    python - <<PY
import json
d=json.load(open("gpurun_out/r03n_bpe_${c}_$kern.json"))
r=d.get("roofline") or {}
print("$kern $c:", d["value"], "MB/s", d["ms_per_step"], "ms", r.get("kernel_us"), r.get("frac"), r.get("dominant_kernel"))
PY
  done
done
