#!/usr/bin/env python3
"""Micro-benchmark of haet_linear_fwd (split-bf16 MFMA GEMM) over a few shapes; prints ms and the
algorithmic HBM rate (rows * (K + O) * 4 B / t)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import ops

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
def _copy_rate():
    a = torch.randn(rows, 256, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"torch copy_ of the same traffic: {ms:.3f} ms  {rows*512*4/ms/1e9:.2f} TB/s")


_copy_rate()
shapes = ((256, 256),) if os.environ.get('PROBE_ONE') else ((256, 256), (256, 128), (128, 256), (512, 256), (256, 512))
for K, O in shapes:
    x = torch.randn(rows, K, device=dev)
    w = ops.SplitLinear(torch.randn(O, K, device=dev) * 0.05, torch.randn(O, device=dev))
    for _ in range(3):
        ops.linear(x, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.linear(x, w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ref = torch.nn.functional.linear(x, torch.randn(O, K, device=dev))
    e0.record()
    for _ in range(5):
        torch.nn.functional.linear(x, torch.randn(O, K, device=dev))
    e1.record()
    torch.cuda.synchronize()
    print(f"K={K:4d} O={O:4d}: {ms:.3f} ms  {rows*(K+O)*4/ms/1e9:.2f} TB/s   "
          f"MFMA {2*3*rows*K*O/ms/1e12:.0f} TF(x3)   [rocBLAS fp32 ~{e0.elapsed_time(e1)/5:.3f} ms]")
