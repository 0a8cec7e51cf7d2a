#!/bin/bash
# Regenerates the measurements kept under profiles/ on a GPU box: bench lines for every configuration, the
# rocprofv3 kernel statistics of the default bench, and the three counter passes.  Each GPU step runs under its
# own time limit; the script stops at the first failure.  Usage (from the repository root): bash tools/collect_profiles.sh
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/final
mkdir -p $OUT
cd $ROOT
for c in c4 c4s c1 c2 c3 c5; do
    timeout -k 10 300 python3 bench.py --config $c > $OUT/${c}_bench.json 2> $OUT/${c}_bench.err
    echo "bench $c done"
done
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29671 \
    bench.py --gpus 2 --steps 5 --warmup 1 --backend gloo --no-idxmap 2> $OUT/gloo2.err | grep '"metric"' > $OUT/c4_gloo2_rehearsal.json
echo "gloo rehearsal done"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29673 \
    bench.py --gpus 2 --steps 5 --warmup 1 --backend gloo --no-idxmap --views-per-gpu 2 2> $OUT/gloo2v2.err | grep '"metric"' > $OUT/c4_gloo2_views2_rehearsal.json
echo "gloo rehearsal with two views per rank done"
timeout -k 10 300 python3 bench.py --no-idxmap --no-cpu-baseline --unfused-train > $OUT/c4_bench_unfused_train.json 2> $OUT/c4_unfused.err
timeout -k 10 300 python3 bench.py --no-idxmap --no-cpu-baseline --views-per-gpu 2 > $OUT/c4_bench_views2.json 2> $OUT/c4_views2.err
echo "training variants done"
timeout -k 10 600 python3 tools/footprint_sweep.py $OUT/c4_footprint.json > $OUT/footprint.log 2>&1
echo "footprint sweep done"
timeout -k 10 600 python3 tools/footprint_sweep.py $OUT/c4s_footprint.json --config c4s > $OUT/footprint_c4s.log 2>&1
echo "footprint sweep of the trained-scene-like workload done"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace done"
# the same with nothing but the forward+backward loop in the process: the per-kernel averages bench.py's roofline figures
# are to be held against (in the full run the training iterations' launches of the same kernels are averaged in)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hot -- python3 $ROOT/bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-idxmap --train-steps 0 --depth-order per_tile_sort > $OUT/trace_hot.log 2>&1
echo "hot-path trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_hot_c4s -- python3 $ROOT/bench.py --config c4s --steps 30 --warmup 5 --no-cpu-baseline --no-idxmap --train-steps 0 --repeats 1 --depth-order gaussians_first > $OUT/trace_hot_c4s.log 2>&1
echo "hot-path trace of the trained-scene-like workload done"
PMC_CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --train-steps 0 --no-idxmap --no-cpu-baseline --depth-order per_tile_sort"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PMC_CMD > $OUT/pmc_write.log 2>&1
echo "write done"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_valu -- $PMC_CMD > $OUT/pmc_valu.log 2>&1
echo "valu done"
# The counter tables above belong to this build: summarise them (tools/install_profiles.sh does the same at home), then
# take the headline lines once more so that their roofline objects quote this build's own traffic and issue figures.
cd $ROOT
if python3 tools/pmc_summary.py --fetch $OUT/pmc_fetch --write $OUT/pmc_write --config c4 --out $ROOT/profiles/hbm_traffic.json --raw $OUT/c4_pmc_traffic.json > $OUT/pmc_summary.log 2>&1 \
   && python3 tools/pmc_valu_summary.py $OUT/pmc_valu $ROOT/profiles/valu_issue.json c4 "rocprofv3 --pmc (collect_profiles.sh)" > $OUT/c4_valu.md 2>> $OUT/pmc_summary.log; then
    cd $ROOT
    timeout -k 10 300 python3 bench.py --config c4 > $OUT/c4_bench.json 2> $OUT/c4_bench.err
    cp $ROOT/profiles/hbm_traffic.json $OUT/hbm_traffic.json
    cp $ROOT/profiles/valu_issue.json $OUT/valu_issue.json
    echo "headline line with this build's counters done"
fi
