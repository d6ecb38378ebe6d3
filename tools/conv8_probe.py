#!/usr/bin/env python3
"""The LDS-DMA convolution (csrc/conv8.hip) against the streaming implicit GEMM of rounds 1-3 (gemm_x3_kernel<CONV>) in
ONE process, interleaved rounds (cdna_hip_programming.md rule 24): C5's per-rank slab (planes x 256 x 256, 256 -> 256
channels, 3 x 3 x 3) by default.  Prints the pre-split pass, both kernels and the bf16-MFMA rate of each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import _lib, ops

dev = torch.device("cuda:0")
planes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
gw = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
Cin = O = 256
grid = (planes, gw, gw)
N = planes * gw * gw
torch.manual_seed(0)
x = torch.randn(1, N, Cin, device=dev)
conv = torch.nn.Conv3d(Cin, O, 3, 1, 1).to(dev)
w = ops.SplitConv(conv.weight, conv.bias)
lib = _lib.load()


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn):
    a, b = ev(), ev()
    a.record()
    r = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b), r


def new_kernel_only(cells, out, f16=False):
    ops.check(lib.haet_conv_project_cells_fwd(ops._p(cells), ops._p(w.f16() if f16 else w.lin.hi), ops._p(None if f16 else w.lin.lo),
                                              ops._p(w.lin.bias), 1, *grid, Cin, O, 3, 3, 3, ops._p(out), int(f16), ops._stream()), "cells")


def old(xx):
    os.environ["HAET_CONV_V1"] = "1"
    try:
        return ops.conv_project(xx, w, grid)
    finally:
        del os.environ["HAET_CONV_V1"]


with torch.no_grad():
    cells = ops.conv_presplit(x)
    out = torch.empty(1, N, O, device=dev)
    new_kernel_only(cells, out)
    ref = old(x)
    err = float((out - ref).norm() / ref.norm())
    variants = [int(v) for v in os.environ.get("C8_VARIANTS", "5").split(",")]
    t = {"presplit": [], "old": [], "new_f16": []}
    cells16 = ops.conv_presplit(x, None, True)
    out16 = torch.empty_like(out)
    new_kernel_only(cells16, out16, True)
    err16 = float((out16 - ref).norm() / ref.norm())
    for v in variants:
        t[f"new{v}"] = []
        os.environ["HAET_C8_VAR"] = str(v)
        new_kernel_only(cells, out)
        e = float((out - ref).norm() / ref.norm())
        assert e < 1e-5, (v, e)
    for _ in range(rounds):
        t["presplit"].append(timed(lambda: ops.conv_presplit(x))[0])
        for v in variants:
            os.environ["HAET_C8_VAR"] = str(v)
            t[f"new{v}"].append(timed(lambda: new_kernel_only(cells, out))[0])
        t["old"].append(timed(lambda: old(x))[0])
        t["new_f16"].append(timed(lambda: new_kernel_only(cells16, out16, True))[0])
fl3 = 3 * 2.0 * N * 27 * Cin * O
med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
print(f"conv {planes}x{gw}x{gw} = {N} points, 256 -> 256, 3x3x3, rel diff new vs old {err:.1e}")
print(f"  presplit {med['presplit']:.3f} ms ({2 * N * Cin * 4 / med['presplit'] / 1e9:.2f} TB/s)")
for k in t:
    if k == "presplit":
        continue
    tf = fl3 * (2.0 / 3.0 if k == "new_f16" else 1.0) / (med[k] * 1e-3) / 1e12
    print(f"  {k:7s} {med[k]:.2f} ms  min {min(t[k]):.2f}  = {tf:.0f} TFLOP/s of bf16 / fp16 MFMA ({tf / 25:.1f} % of 2.5 PF)")
print(f"  fp16x2 vs bf16x3 output: rel diff {err16:.1e}")
