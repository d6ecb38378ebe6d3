#!/bin/bash
# Stamped diagnostic build ON THE GPU BOX for BASELINE config C2's token stage (H = 4, dim_head 64, M = 32: erwin.hip)
make -C haet_amd/csrc clean >/dev/null
make -C haet_amd/csrc -j16 EXTRA=-DHAET_STAMPS > gpurun_out/build_stamps.log 2>&1 || { tail -5 gpurun_out/build_stamps.log; exit 1; }
STAMP_SHAPE=4,64,32 NO_FLUSH=1 python tools/erwin_stamps.py > gpurun_out/erwin_stamps_c2.txt 2>&1
make -C haet_amd/csrc clean > /dev/null
make -C haet_amd/csrc -j16 EXTRA="-DHAET_STAMPS -DHAET_STAMP_LVL=1" >> gpurun_out/build_stamps.log 2>&1
STAMP_SHAPE=4,64,32 NO_FLUSH=1 python tools/erwin_stamps.py > gpurun_out/erwin_stamps_c2_l1.txt 2>&1
echo "== C2 level 0"; cat gpurun_out/erwin_stamps_c2.txt; echo "== C2 level 1"; tail -14 gpurun_out/erwin_stamps_c2_l1.txt
