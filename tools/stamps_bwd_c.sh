#!/bin/bash
make -C haet_amd/csrc clean >/dev/null
make -C haet_amd/csrc -j16 EXTRA="-DHAET_STAMPS -DHAET_STAMP_PHASE_C" > gpurun_out/build_stamps.log 2>&1 || { tail -5 gpurun_out/build_stamps.log; exit 1; }
STAMP_PHASE_C=1 python tools/erwin_bwd_stamps.py
