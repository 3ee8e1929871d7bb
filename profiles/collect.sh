#!/bin/bash
# Collects the round's evidence on the MI355X box (run from the repo root through gpurun):
#   bash profiles/collect.sh <tag> [config]   -> gpurun_out/<tag>_*   (copy the summaries you keep into profiles/)
# 1. bench.py with the CPU baseline (the JSON line)           2. rocprofv3 --kernel-trace --stats of the same command
# 3./4. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (no trace domains besides --kernel-trace)
# 5. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE: matrix-core utilisation per kernel
#    = MFMA busy cycles (summed over the SIMDs) / (128 SIMDs per XCD x GRBM_GUI_ACTIVE, which rocprofv3 reports summed over the 8 XCDs)
# The summary <tag>_pmc_hbm.json is what bench.py reads as profiles/traffic_<config>.json.
# The conv plans tuned in step 1 are reloaded (ARSEG_CONV_PLAN_FILE) so that the profiled runs contain no trial launches.
TAG=${1:-r01}
CFG=${2:-psp}          # bench.py --config (psp = the headline workload)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export ARSEG_CONV_PLAN_FILE=$OUT/${TAG}_plans.json
rm -f $ARSEG_CONV_PLAN_FILE
cd $R
python bench.py --config $CFG --steps 9 --warmup 3 --conv-layers $OUT/${TAG}_conv_layers.json 2> $OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline --no-variants"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o s --output-format csv -- $BENCH > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_fetch -o f --output-format csv -- $BENCH --no-profile > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_write -o w --output-format csv -- $BENCH --no-profile > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE -d $OUT/${TAG}_mfma -o m --output-format csv -- $BENCH --no-profile > /dev/null 2>&1
python $R/tools/summarize_profile.py $OUT $TAG
# the raw traces are tens of MB each; gpurun merges at most 64 MiB back: keep the summaries only (KEEP_RAW=1 keeps everything)
[ -n "$KEEP_RAW" ] || rm -rf $OUT/${TAG}_stats $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_mfma
