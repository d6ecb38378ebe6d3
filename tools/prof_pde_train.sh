#!/bin/bash
# kernel summary + host profile of tools/pde_configs_probe.py for ONE configuration (PDE_ONLY=substring) ON THE GPU BOX
export TMPDIR=/tmp; R=$(pwd); tag=${2:-pde_train}; mkdir -p $R/gpurun_out/$tag; cd /tmp
export PDE_ONLY="$1"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/pde_configs_probe.py > $R/gpurun_out/$tag/log.txt 2>&1
cd $R
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows); n=sum(int(r["Calls"]) for r in rows)
print("total kernel time %.1f ms in %d launches" % (tot/1e6, n))
for r in rows[:28]:
    print("%9.1f us x %6s  %5.1f%%  %s" % (float(r["AverageNs"])/1e3, r["Calls"], float(r["Percentage"]), r["Name"][:120]))
PY
python3 -c "
import cProfile,pstats,runpy,sys,io
pr=cProfile.Profile(); pr.enable()
runpy.run_path('tools/pde_configs_probe.py', run_name='__main__')
pr.disable(); s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
" > gpurun_out/$tag/host_profile.txt 2>&1
