# round 2, last call: the whole GPU suite, smoke, the default bench line, and its rocprofv3 kernel summary (profiles/r02d_*)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/gpu_r02_full.sh || exit 1
cd /tmp
O=$R/gpurun_out/prof_r02d
rm -rf $O; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline -- python3 $R/bench.py --steps 20 --warmup 5 > $O/headline.json 2> $O/headline.err; echo "headline trace exit=$?"
f=$(find $O/headline -name "*kernel_stats.csv" | head -1); cp "$f" $O/headline_kernel_stats.csv; cut -d, -f1-4 "$f" | cut -c1-120 | head -5
rm -rf $O/headline
