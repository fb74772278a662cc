#!/bin/bash
# Collects the round's judged artefacts on the GPU box into gpurun_out/<tag>/ (copy the summaries into profiles/ afterwards).
# usage: bash tools/collect_profiles.sh <tag>      (every rocprofv3 run has python3 itself after "--"; PMC passes are separate runs)
set -e -o pipefail
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-stack"
PMCB="bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-stack"
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_full_line.json 2> $OUT/bench_full_line.err
echo "bench done"
P2I_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o serial -- python3 $BENCH > $OUT/serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/side -o side -- python3 $BENCH > $OUT/side.log 2>&1
echo "kernel stats done"
P2I_SIDE_WGRAD=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $PMCB > $OUT/fetch.log 2>&1
P2I_SIDE_WGRAD=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $PMCB > $OUT/write.log 2>&1
echo "traffic done"
P2I_SIDE_WGRAD=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/mfma -o mfma -- python3 $PMCB > $OUT/mfma.log 2>&1
P2I_SIDE_WGRAD=0 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/lds -o lds -- python3 $PMCB > $OUT/lds.log 2>&1
echo "pmc done"
python3 tools/pmc_traffic.py $OUT/fetch/fetch_counter_collection.csv $OUT/write/write_counter_collection.csv $OUT/pmc_hbm_traffic.csv $OUT/pmc_traffic.json \
  "profiles/${TAG}_pmc_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of: P2I_SIDE_WGRAD=0 python3 $PMCB)"
python3 tools/pmc_summary.py $OUT/mfma/mfma_counter_collection.csv $OUT/lds/lds_counter_collection.csv > $OUT/pmc_mfma_busy.csv
rm -f $OUT/*/*kernel_trace.csv $OUT/*/*counter_collection.csv $OUT/*/*agent_info.csv
ls -la $OUT
