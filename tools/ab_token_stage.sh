#!/bin/bash
# A/B of the fused token stage against the three-launch path, 1M points and the 8-GPU shard size (same process order twice)
set -e
for i in 1 2; do
  python bench.py --steps 30 --no-cpu-baseline > gpurun_out/b_fused_$i.json
  HAET_NO_TOKEN_STAGE=1 python bench.py --steps 30 --no-cpu-baseline > gpurun_out/b_unfused_$i.json
done
python bench.py --steps 50 --points 125000 --no-cpu-baseline > gpurun_out/b125_fused.json
HAET_NO_TOKEN_STAGE=1 python bench.py --steps 50 --points 125000 --no-cpu-baseline > gpurun_out/b125_unfused.json
python tools/bench_lines.py gpurun_out/b_fused_*.json gpurun_out/b_unfused_*.json gpurun_out/b125_*.json
