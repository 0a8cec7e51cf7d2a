set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_now
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-idxmap --train-steps 0 > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kernel_stats_md.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) > $OUT/kernel_stats.md 2>/dev/null || true
cat $OUT/kernel_stats.md | head -30
