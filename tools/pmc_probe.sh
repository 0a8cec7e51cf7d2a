set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcx
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --train-steps 0 --no-idxmap --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
echo a done
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INSTS_SALU --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
echo b done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_dump.py $OUT/a $OUT/b render_bwd render_fwd > $OUT/summary.txt; cat $OUT/summary.txt
