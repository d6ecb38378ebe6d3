#!/bin/bash
# Stamped diagnostic builds ON THE GPU BOX (the snapshot's library is rebuilt there; nothing comes back but text)
make -C haet_amd/csrc clean >/dev/null
make -C haet_amd/csrc -j16 EXTRA=-DHAET_STAMPS > gpurun_out/build_stamps.log 2>&1 || { tail -5 gpurun_out/build_stamps.log; exit 1; }
python tools/erwin_stamps.py > gpurun_out/erwin_stamps.txt 2>&1
NO_FLUSH=1 python tools/erwin_stamps.py > gpurun_out/erwin_stamps_noflush.txt 2>&1
HAET_NO_TOKEN_STAGE=1 python tools/erwin_stamps.py > gpurun_out/erwin_stamps_unfused.txt 2>&1
make -C haet_amd/csrc clean > /dev/null
make -C haet_amd/csrc -j16 EXTRA="-DHAET_STAMPS -DHAET_STAMP_LVL=1" >> gpurun_out/build_stamps.log 2>&1
python tools/erwin_stamps.py > gpurun_out/erwin_stamps_l1.txt 2>&1
echo "== fused"; cat gpurun_out/erwin_stamps.txt; echo "== fused, no cache flush"; cat gpurun_out/erwin_stamps_noflush.txt; echo "== unfused"; head -14 gpurun_out/erwin_stamps_unfused.txt; echo "== level 1"; tail -12 gpurun_out/erwin_stamps_l1.txt
