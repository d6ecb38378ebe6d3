#!/bin/bash
# Ablation of the fused de-slice + to_out kernel (HAET_DO_DBG bits, see deslice_out.hip): stage time per variant.
# The switches exist in the stamped build only:  touch haet_amd/csrc/deslice_out.hip && make -C haet_amd/csrc EXTRA=-DHAET_STAMPS
# (on the GPU box; rebuild without EXTRA afterwards).  tools/deslice_out_stamps.py prints the per-phase cycles of the same build.
for d in 0 4 8 1 2 3; do
  HAET_DO_DBG=$d python bench.py --steps 20 --no-cpu-baseline 2>/dev/null > gpurun_out/abl_$d.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/abl_$d.json").read().strip().splitlines()[-1])
print("dbg=%2d  deslice_out %.3f ms  (forward %.3f)" % ($d, d["stage_ms"]["deslice"], d["ms_per_step"]))
PY
done
