#!/bin/bash
# The measurements quoted in DESIGN.md for a build, in one GPU call (outputs under gpurun_out/round/)
mkdir -p gpurun_out/round; O=gpurun_out/round
python bench.py --steps 30 > $O/bench_1m.json 2> $O/bench_1m.err
python bench.py --steps 50 --points 125000 --no-cpu-baseline > $O/bench_125k.json 2>> $O/bench_1m.err
HAET_NO_TOKEN_STAGE=1 HAET_NO_DESLICE_OUT=1 python bench.py --steps 30 --no-cpu-baseline > $O/bench_1m_unfused.json 2>> $O/bench_1m.err
python bench.py --sweep --steps 5 > $O/sweep.json 2> $O/sweep.txt
python tools/train_probe.py > $O/train_probe.txt 2>&1
python tools/model_probe.py > $O/model_probe.txt 2>&1
python tools/model_probe.py --train > $O/model_train.txt 2>&1
python tools/c1_probe.py > $O/c1.txt 2>&1
python tools/shape_probe.py > $O/c2.txt 2>&1
python tools/c5_probe.py > $O/c5.txt 2>&1
python tools/bench_lines.py $O/bench_1m.json $O/bench_125k.json $O/bench_1m_unfused.json
tail -2 $O/train_probe.txt; tail -1 $O/model_probe.txt; tail -1 $O/model_train.txt; tail -2 $O/c1.txt; tail -3 $O/c2.txt; tail -1 $O/c5.txt; cat $O/sweep.txt
python tools/pc_probe.py --cpu > $O/pointcloud_32k.json 2> $O/pointcloud.err
python tools/pc_probe.py --points 1000000 --radius 0.012 --steps 3 > $O/pointcloud_1m.json 2>> $O/pointcloud.err
cat $O/pointcloud_32k.json $O/pointcloud_1m.json
