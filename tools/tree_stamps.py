#!/usr/bin/env python3
"""Cycle stamps along the ball-tree build (wave 0 of cloud 0).  Needs the stamped build:
make -C haet_amd/csrc clean all EXTRA=-DHAET_STAMPS   (GPU box only; never quote its run time)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import _lib, ops

dev = torch.device("cuda:0")
S = torch.randn(512, 32, device=dev)
for _ in range(3):
    ops.balltree_build(None, 8, 64, [2], [32, 16], 45.0, tokens=S, d=3)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 32)()
lib.haet_debug_read_tree_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.haet_debug_read_tree_stamps(buf) == 0
t = list(buf)
print("token positions:", t[1] - t[0], "cycles")
print("main tree levels (64, 32, 16, 8, 4):", [t[2 + i] - t[1 + i] for i in range(5)])
print("  inside each main level [keys, selects, exchange + barrier]:", [[t[16 + 2 * i] - t[1 + i], t[17 + 2 * i] - t[16 + 2 * i], t[2 + i] - t[17 + 2 * i]] for i in range(5)])
print("to the start of the rotated re-partitions:", t[10] - t[6])
print("rotated levels (two hierarchy levels at once):", [t[11 + i] - t[10 + i] for i in range(4)])
print("total:", t[14] - t[0], "cycles")
