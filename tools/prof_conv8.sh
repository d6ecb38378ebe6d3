#!/bin/bash
# rocprofv3 passes of the LDS-DMA convolution at C5's slab ON THE GPU BOX (kernel-trace stats, then the SQ / GRBM counters that
# give matrix-pipe utilisation and the clock held; one counter group per run, no trace domains mixed with --pmc).
#   usage: bash tools/prof_conv8.sh [tag]   -> gpurun_out/prof_conv8_<tag>/summary.txt
TAG=${1:-r04}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_conv8_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/conv8_probe.py 32 256 3 > $OUT/stats_probe.txt 2> $OUT/stats.log
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/tools/conv8_probe.py 32 256 1 > $OUT/pmc_$name.txt 2> $OUT/pmc_$name.log || echo "pass $name failed (see $OUT/pmc_$name.log)"
done
cd $ROOT
python3 tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/stats_probe.txt; tail -30 $OUT/summary.txt
