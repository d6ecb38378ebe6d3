#!/usr/bin/env python3
"""Cycle stamps along one row tile of the W-stationary GEMM (waves 0 and 4 of workgroup 5, tile 10).
Needs the stamped build:  make -C haet_amd/csrc clean all EXTRA=-DHAET_STAMPS   (GPU box only;
never quote the run time of that build)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import _lib, ops

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = torch.randn(rows, 256, device=dev)
w = ops.SplitLinear(torch.randn(256, 256, device=dev) * 0.05, torch.randn(256, device=dev))
for _ in range(3):
    ops.linear(x, w)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 64)()
lib.haet_debug_read_gemm_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.haet_debug_read_gemm_stamps(buf) == 0
for wv in (0, 1):
    t = [buf[32 * wv + i] for i in range(11)]
    print(f"wave {4 * wv}: cycles from tile start to [12 MFMAs of k-item kt issued] for kt = 0..7, then to after the barrier")
    print("   ", [t[i] - t[0] for i in range(1, 10)])
    print("    increments", [t[i] - t[i - 1] for i in range(1, 10)], " stores of the tile issued after", t[10] - t[9])
