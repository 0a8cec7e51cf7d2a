#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/final into profiles/ and summarises the counter passes.
# Usage: bash tools/install_profiles.sh r01
set -e
TAG=${1:-r01}
SRC=gpurun_out/final
# results of earlier collections pile up next to the newest (every run's files carry its process id): keep the newest
for d in "$SRC/trace" "$SRC/trace_hot" "$SRC/trace_hot_c4s" "$SRC/pmc_fetch" "$SRC/pmc_write" "$SRC/pmc_valu"; do
    [ -d "$d" ] || continue
    for sub in "$d"/*/; do
        [ -d "$sub" ] || continue
        newest=$(ls -t "$sub" | head -1 | sed 's/_.*//')
        [ -n "$newest" ] || continue
        find "$sub" -type f ! -name "${newest}_*" -delete
    done
done
for c in c1 c2 c3 c4 c4s c5; do cp $SRC/${c}_bench.json profiles/${TAG}_${c}_bench.json; done
cp $SRC/c4_gloo2_rehearsal.json profiles/${TAG}_c4_gloo2_rehearsal.json
for f in c4_gloo2_views2_rehearsal c4_bench_unfused_train c4_bench_views2 c4_footprint c4s_footprint; do
    if [ -f $SRC/$f.json ]; then cp $SRC/$f.json profiles/${TAG}_$f.json; fi
done
cp $SRC/trace/*/*_kernel_stats.csv profiles/${TAG}_c4_final_kernel_stats.csv
python3 tools/kernel_stats_md.py profiles/${TAG}_c4_final_kernel_stats.csv "python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline" > profiles/${TAG}_c4_final_kernel_stats.md
if [ -d $SRC/trace_hot ]; then
    cp $SRC/trace_hot/*/*_kernel_stats.csv profiles/${TAG}_c4_hot_path_kernel_stats.csv
    python3 tools/kernel_stats_md.py profiles/${TAG}_c4_hot_path_kernel_stats.csv "python3 bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-idxmap --train-steps 0 --depth-order per_tile_sort" > profiles/${TAG}_c4_hot_path_kernel_stats.md
fi
if [ -d $SRC/trace_hot_c4s ]; then
    cp $SRC/trace_hot_c4s/*/*_kernel_stats.csv profiles/${TAG}_c4s_hot_path_kernel_stats.csv
    python3 tools/kernel_stats_md.py profiles/${TAG}_c4s_hot_path_kernel_stats.csv "python3 bench.py --config c4s --steps 30 --warmup 5 --no-cpu-baseline --no-idxmap --train-steps 0 --repeats 1 --depth-order gaussians_first" > profiles/${TAG}_c4s_hot_path_kernel_stats.md
fi
python3 tools/pmc_summary.py --fetch $SRC/pmc_fetch --write $SRC/pmc_write --config c4 \
    --out profiles/hbm_traffic.json --raw profiles/${TAG}_c4_pmc_traffic.json
python3 tools/pmc_valu_summary.py $SRC/pmc_valu profiles/valu_issue.json c4 "rocprofv3 --pmc SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" > profiles/${TAG}_c4_valu.md
python3 tools/results_tables.py $TAG
