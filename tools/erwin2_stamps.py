#!/usr/bin/env python3
"""Diagnostic: where does one cloud of the fragment kernel (csrc/erwin2.hip) spend its cycles?
Needs the stamped build of that file:  make -C haet_amd/csrc EXTRA=-DHAET_STAMPS  after touching erwin2.hip (GPU box
only; rebuild without EXTRA afterwards).  Prints s_memtime deltas between op boundaries of cloud 0, the phases of the
op chosen on the command line (default 0 = first block), and the shader clock from s_memrealtime (100 MHz)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import _lib
from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh

dev = torch.device("cuda:0")
torch.manual_seed(0)
mod = Physics_Attention_Irregular_Mesh(256, heads=8, dim_head=32, slice_num=64, rotate=45).eval().to(dev)
x = torch.randn(1, 200_000, 256, device=dev)
big = torch.empty(256 * 1024 * 1024, device=dev)  # flush caches between forwards like the 1M-point run
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 128)()
ops = [int(a) for a in sys.argv[1:]] or [0]
for stamp_op in ops:
    lib.haet_debug_read_stamps2(buf, stamp_op)  # (sets the op to stamp)
    with torch.no_grad():
        for _ in range(3):
            if not os.environ.get("NO_FLUSH"):
                big.fill_(1.0)
            mod(x)
    torch.cuda.synchronize()
    assert lib.haet_debug_read_stamps2(buf, stamp_op) == 0
    t = list(buf)
    clk = (t[20] - t[0]) / max(1, (t[64 + 20] - t[64 + 0])) * 100.0
    print(f"shader clock during the kernel: {clk:.0f} MHz; total {t[20]-t[0]} cycles = {(t[64+20]-t[64])/100.0:.1f} us")
    names = ["tree build", "prologue (gather, maps, rel)"] + [f"op {i}" for i in range(16)]
    prev = 0
    for i in range(1, 19):
        if t[i] == 0 or t[i] < t[prev]:
            continue
        print(f"  {names[i-1]:30s} +{t[i]-t[prev]:8d} cycles")
        prev = i
    print(f"  {'scatter out':30s} +{t[20]-t[prev]:8d} cycles")
    an = {33: "attn: weights issued + load_x", 34: "attn: mid barrier", 35: "attn: norm + PE + split", 36: "attn: qk / v MFMAs, S",
          37: "attn: softmax", 38: "attn: PV + proj", 39: "attn: slab store + barrier", 40: "mlp: weights issued + load", 41: "mlp: mid barrier",
          42: "mlp: store X, stage params, norm, split", 43: "mlp: w1 / w2 MFMAs", 44: "mlp: SwiGLU gate", 45: "mlp: w3 MFMAs",
          46: "mlp: slab store + barrier"}
    print(f"phases of op {stamp_op} (wave 0):")
    for i in range(33, 47):
        print(f"  {an[i]:40s} {t[i]-t[i-1]:8d} cycles")

tb = (ctypes.c_ulonglong * 32)()
if lib.haet_debug_read_tree_stamps2(tb) == 0:
    t = list(tb)
    print("builder inside the fused kernel: token positions", t[1] - t[0], "main levels", [t[2 + i] - t[1 + i] for i in range(5)],
          "to rotated", t[10] - t[6], "rotated levels", [t[11 + i] - t[10 + i] for i in range(4)], "total", t[14] - t[0])
