#!/bin/bash
make -C haet_amd/csrc clean >/dev/null
make -C haet_amd/csrc -j16 EXTRA=-DHAET_STAMPS > gpurun_out/build_stamps.log 2>&1 || { tail -5 gpurun_out/build_stamps.log; exit 1; }
python tools/erwin_bwd_stamps.py
