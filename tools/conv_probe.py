#!/usr/bin/env python3
"""haet_conv_project_fwd (implicit GEMM, split-bf16 MFMA) against MIOpen's fp32 Conv3d on a channels-
last 3-D grid: the in_project_x of Physics_Attention_Structured_Mesh_3D (C5 is 256^3; here 96^3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import ops

dev = torch.device("cuda:0")
g = int(sys.argv[1]) if len(sys.argv) > 1 else 96
Cin = O = 256
x = torch.randn(1, g * g * g, Cin, device=dev)
conv = torch.nn.Conv3d(Cin, O, 3, 1, 1).to(dev)
w = ops.SplitConv(conv.weight, conv.bias)


def timeit(fn, n=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def miopen():
    xg = x.reshape(1, g, g, g, Cin).permute(0, 4, 1, 2, 3).contiguous()
    return conv(xg).permute(0, 2, 3, 4, 1).reshape(1, -1, O)


with torch.no_grad():
    t_hip = timeit(lambda: ops.conv_project(x, w, (g, g, g)))
    t_mi = timeit(miopen)
    err = ((ops.conv_project(x, w, (g, g, g)) - miopen()).norm() / miopen().norm()).item()
fl = 2.0 * g ** 3 * 27 * Cin * O
print(f"grid {g}^3, 256 -> 256 channels, 3x3x3: HIP implicit GEMM {t_hip:.2f} ms ({fl*3/t_hip/1e9:.0f} TFLOP/s of bf16 MFMA, "
      f"{fl/t_hip/1e9:.0f} fp32-equivalent); MIOpen fp32 incl. layout permutes {t_mi:.2f} ms; rel diff {err:.1e}")
