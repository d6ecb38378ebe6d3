#!/usr/bin/env python3
"""Diagnostic: where do the waves of the fused de-slice + to_out kernel (csrc/deslice_out.hip) spend their cycles?
Needs the stamped build of that file:  touch haet_amd/csrc/deslice_out.hip && make -C haet_amd/csrc EXTRA=-DHAET_STAMPS
(GPU box only; rebuild without EXTRA afterwards).  Prints, per wave of workgroup 0, s_memtime cycles per tile in each
part of the loop.  HAET_DO_DBG (see the kernel) applies: 4 = no stagger."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_module
from haet_amd import _lib

dev = torch.device("cuda:0")
mod = build_module(dev).eval()
x = torch.randn(1, int(os.environ.get("POINTS", 1_000_000)), 256, device=dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
with torch.no_grad():
    for _ in range(3):
        mod(x)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert lib.haet_debug_read_do_stamps(buf) == 0
t = list(buf)
names = ["stage", "[A] de-slice", "[B] to_out", "fetch + touch", "barrier"]
print("s_memtime ticks per tile (100 MHz constant clock x ?; compare columns, not absolute)")
print("wave  tiles " + " ".join(f"{n:>14s}" for n in names) + "          total")
for w in range(8):
    n = max(1, t[w * 8 + 5])
    v = [t[w * 8 + i] / n for i in range(5)]
    print(f"{w:4d} {n:6d} " + " ".join(f"{a:14.1f}" for a in v) + f" {sum(v):14.1f}")
