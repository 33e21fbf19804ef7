export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_tr
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tr
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tr -- python3 bench.py --workload bpe_train --corpus ${1:-lex} --steps 1 > gpurun_out/prof_tr.json 2> gpurun_out/prof_tr.err; echo prof_exit=$?
f=$(find gpurun_out/prof_tr -name "*kernel_stats.csv" | head -1)
cut -d, -f1-8 $f | cut -c1-200 | head -30
t=$(find gpurun_out/prof_tr -name "*kernel_trace.csv" | head -1)
python tools/analyze_trace.py $t fast_apply_kernel,fast_tie_kernel | tee gpurun_out/prof_tr_analysis.txt
rm -f $t   # large; the analysis is what travels back
python - <<'PY'
import json
d=json.load(open("gpurun_out/prof_tr.json")); print(d["value"], d["roofline"]["kernel_us"], d["config"]["workload"])
PY
