#!/bin/bash
# NOTE: this script measured a form that did not stay in the tree (see profiles/r03_experiments/); its build flag / environment knob exists only in the commit it ran against.
# bpe_lane_kernel with the fused output (tile_lookback) and the packed-key scan: BPE parity tests, then bench lines
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "bpe_ or smoke or cli or dedup or headline_corpus_encode or single_launch or random_tables or lowercase or joined or config4_mixed" > gpurun_out/r03q_pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r03q_pytest.log
if [ $rc -ne 0 ]; then head -80 gpurun_out/r03q_pytest.log; exit $rc; fi
one() {  # label, corpus
  timeout -k 10 300 python bench.py --workload bpe_encode --corpus $2 --lean --steps 100 --warmup 10 > gpurun_out/r03q_$1.json 2> gpurun_out/r03q_$1.err || { tail -5 gpurun_out/r03q_$1.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03q_$1.json"))
print("$1 $2:", d["value"], "MB/s", d["ms_per_step"], "ms", flush=True)
PY
}
one fused open && SWT_BPE_FUSED=0 one scan open && one lex_u128 lex && SWT_BPE_UTILE=256 one lex_u256 lex && SWT_BPE_UTILE=64 one lex_u64 lex
cd /tmp && rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --workload bpe_encode --corpus open --lean --steps 20 --warmup 3 > /tmp/kt.log 2>&1; f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r03q_open_kernel_stats.csv; head -8 $f | cut -c1-200
