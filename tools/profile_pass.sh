#!/bin/bash
# Profiling passes of the headline bench ON THE GPU BOX (rocprofv3; one counter group per run, no trace domains mixed
# with --pmc).  Everything lands under gpurun_out/prof_<tag>/; tools/profile_summary.py turns it into profiles/ files.
#   usage: bash tools/profile_pass.sh <tag> [extra bench args]
TAG=${1:-r02}; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-train-leg $@"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-leg $@ > $OUT/stats_bench.json 2> $OUT/stats.log
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$name.json 2> $OUT/pmc_$name.log || echo "pass $name failed (see $OUT/pmc_$name.log)"
done
cd $ROOT
python3 tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
