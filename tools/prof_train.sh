#!/bin/bash
# kernel summary of the training-step probe (GPU box)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_train; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/train_probe.py > $OUT/probe.txt 2> $OUT/log.txt
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{float(r['AverageNs'])/1e3:10.1f} us x {r['Calls']:>5}  {float(r['Percentage']):5.1f}%  {r['Name'][:90]}")
PY
