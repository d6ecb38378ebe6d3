#!/usr/bin/env python3
"""The weight-gradient kernels (csrc/gemm_tn.hip) at the PDE-benchmark sizes and at 1M rows: a Linear's G^T.X and the nine
taps of a 3 x 3 convolution; HAET_WGRAD_CHUNK (rows per workgroup, read once per process) is the sweep knob."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from haet_amd import ops

dev = torch.device("cuda", 0)


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - a) / n * 1e6


for rows, O, K, grid in [(4 * 11271, 128, 128, (221, 51)), (8 * 16641, 128, 128, (129, 129)), (125000, 256, 256, None), (1000000, 256, 256, None)]:
    g, x = torch.randn(rows, O, device=dev), torch.randn(rows, K, device=dev)
    lin = t(lambda: ops.linear_wgrad(g, x))
    conv = None
    if grid is not None:
        B = rows // (grid[0] * grid[1])
        conv = t(lambda: ops.conv_wgrad(g.view(B, -1, O), x.view(B, -1, K), grid, (3, 3)))
    print(f"chunk={os.environ.get('HAET_WGRAD_CHUNK', 'auto')} rows={rows} {O}x{K}: linear wgrad {lin:.0f} us" + ("" if conv is None else f", 3x3 conv wgrad {conv:.0f} us"), flush=True)
