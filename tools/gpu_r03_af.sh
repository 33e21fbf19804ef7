#!/bin/bash
# wordref_kernel: class table packed into 256 bytes of LDS (6,368 -> 5,600 B) and registers capped for 7 waves per SIMD
set -o pipefail
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "wp_ or dedup or config2 or config4 or smoke or word_lane or lowercase" > gpurun_out/r03af_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03af_pytest.log
if [ $rc -ne 0 ]; then head -80 gpurun_out/r03af_pytest.log; exit $rc; fi
one() {  # label, args
  local label=$1; shift
  timeout -k 10 400 python bench.py "$@" --lean > gpurun_out/r03af.json 2> gpurun_out/r03af.err || { tail -5 gpurun_out/r03af.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03af.json"))
print("$label:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
}
for v in "-DSWT_WORDREF_WAVES=7" "-DSWT_WORDREF_WAVES=6"; do
  export SWT_EXTRA_FLAGS="$v"
  python -c "import importlib; importlib.import_module('subword-tokenizers_amd._build').build()" || exit 1
  one "[$v] wp" --workload wp_encode --steps 20 --warmup 5
  one "[$v] mixed" --workload mixed_encode --steps 10 --warmup 3
  SWT_BPE_DEDUP=2 one "[$v] lex (dedup forced)" --workload bpe_encode --corpus lex --steps 50 --warmup 10
done
