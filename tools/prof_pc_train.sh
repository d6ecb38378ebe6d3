#!/bin/bash
# kernel summary of the stand-alone model's training step (GPU box); args go to tools/pc_train_probe.py
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_pc_train; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/pc_train_probe.py "$@" > $OUT/probe.txt 2> $OUT/log.txt
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print(f"{float(r['AverageNs'])/1e3:10.1f} us x {r['Calls']:>5}  {float(r['Percentage']):5.1f}%  {r['Name'][:110]}")
PY
tail -1 $OUT/probe.txt
