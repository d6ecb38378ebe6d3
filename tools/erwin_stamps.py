#!/usr/bin/env python3
"""Diagnostic: where does one cloud of the Erwin kernel spend its cycles?
Needs the stamped build:  make -C haet_amd/csrc clean all EXTRA=-DHAET_STAMPS   (GPU box only;
rebuild without EXTRA afterwards).  Prints s_memtime deltas between phase boundaries of cloud 0
and the shader clock derived from s_memrealtime (100 MHz)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import _lib
from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh

dev = torch.device("cuda:0")
torch.manual_seed(0)
# shape: STAMP_SHAPE="heads,dim_head,slice_num" (default: config C3; C2 = "4,64,32")
_h, _c, _m = (int(v) for v in os.environ.get("STAMP_SHAPE", "8,32,64").split(","))
mod = Physics_Attention_Irregular_Mesh(256, heads=_h, dim_head=_c, slice_num=_m, rotate=45).eval().to(dev)
x = torch.randn(1, 200_000, 256, device=dev)
big = torch.empty(256 * 1024 * 1024, device=dev)  # flush caches between forwards like the 1M-point run
with torch.no_grad():
    for _ in range(3):
        if not os.environ.get("NO_FLUSH"):
            big.fill_(1.0)
        mod(x)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 128)()
assert lib.haet_debug_read_stamps(buf) == 0
t = list(buf)
clk = (t[12] - t[0]) / max(1, (t[64 + 12] - t[64 + 0])) * 100.0
print(f"shader clock during the kernel: {clk:.0f} MHz; total {t[12]-t[0]} cycles = {(t[64+12]-t[64])/100.0:.1f} us")
names = {1: "prologue", 2: "gather to tree order", 3: "enc0 block 0", 4: "enc0 block 1", 5: "pool",
         6: "bottleneck block 0", 7: "bottleneck block 1", 8: "unpool", 9: "dec0 block 0",
         10: "dec0 block 1", 12: "scatter out"}
prev = 0
for i in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12):
    print(f"  {names[i]:22s} +{t[i]-t[prev]:8d} cycles")
    prev = i
print("phases of the last block recorded (dec0 block 1; with EXTRA='-DHAET_STAMPS -DHAET_STAMP_LVL=1': bottleneck block 1):")
bn = {17: "params+rel_pos", 18: "norm1+pe", 19: "qkv", 20: "attention", 21: "proj", 22: "norm2", 23: "w1w2", 24: "w3"}
for i in range(17, 25):
    print(f"  {bn[i]:14s} {t[i]-t[i-1]:8d} cycles")
print("inside the last proj/w3-type Linear call (wave 0): entry -> fetch issued -> operands read -> MFMAs done -> exit")
print("  ", [t[i] - t[40] for i in range(40, 45)], " (phase entry stamp to linear entry:", t[40] - t[23], ")")
