#!/bin/bash
# kernel summary of tools/pde_shape_probe.py (the reference's PDE-benchmark widths) ON THE GPU BOX
export TMPDIR=/tmp; R=$(pwd); mkdir -p $R/gpurun_out/prof_pde; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pde -- python3 $R/tools/pde_shape_probe.py > $R/gpurun_out/prof_pde/log.txt 2>&1
cd $R
f=$(find gpurun_out/prof_pde -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%9.1f us x %5s  %5.1f%%  %s" % (float(r["AverageNs"])/1e3, r["Calls"], float(r["Percentage"]), r["Name"][:110]))
PY
