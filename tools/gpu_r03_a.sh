#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite (new: headline corpus + configs[3] at spec), then the default bench line
# (now with wp_encode / mixed_encode blocks) and the WordPiece training line.  A step that is KILLED ends the call.
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r03a_pytest.log 2>&1
rc=$?
tail -5 gpurun_out/r03a_pytest.log
if [ $rc -ge 124 ]; then echo "pytest killed (rc $rc)"; exit $rc; fi
timeout -k 10 500 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
rc2=$?
echo "bench rc $rc2"; tail -3 gpurun_out/r03a_bench.err; head -c 600 gpurun_out/r03a_bench.json
if [ $rc2 -ge 124 ]; then exit $rc2; fi
timeout -k 10 200 python bench.py --workload wp_train > gpurun_out/r03a_wp_train.json 2> gpurun_out/r03a_wp_train.err
echo "wp_train rc $?"; head -c 400 gpurun_out/r03a_wp_train.json
exit $rc
