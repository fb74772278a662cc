set -e -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PMCB="bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-stack"
P2I_SIDE_WGRAD=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/mfma -o mfma -- python3 $PMCB > $OUT/mfma.log 2>&1
P2I_SIDE_WGRAD=0 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $OUT/wait -o wait -- python3 $PMCB > $OUT/wait.log 2>&1
python3 tools/pmc_summary.py $OUT/mfma/mfma_counter_collection.csv $OUT/wait/wait_counter_collection.csv > $OUT/pmc_x6p.csv
rm -f $OUT/*/*counter_collection.csv $OUT/*/*agent_info.csv
