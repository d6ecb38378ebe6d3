#!/usr/bin/env python3
"""Where the LDS-DMA convolution (pre-split pass included) overtakes the streaming kernel: a sweep over grid sizes at 256 -> 256
channels, one process, interleaved (sets ops.CONV_CELLS_MIN_ROWS)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fn, n=5):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[n // 2]


for grid in [(64, 64), (128, 128), (181, 181), (256, 256), (512, 512), (16, 16, 16), (24, 24, 24), (32, 32, 32), (48, 48, 48), (64, 64, 64)]:
    N = 1
    for g in grid:
        N *= g
    x = torch.randn(1, N, 256, device=dev)
    W = torch.randn(256, 256, *([3] * len(grid)), device=dev) * 0.02
    w = ops.SplitConv(W, torch.zeros(256, device=dev))
    with torch.no_grad():
        os.environ["HAET_CONV_V1"] = "1"
        t_old = timed(lambda: ops.conv_project(x, w, grid))
        del os.environ["HAET_CONV_V1"]
        t_new = timed(lambda: ops.conv_project_cells(x, w, grid))
    print(f"grid {grid} = {N} rows ({(N + 255) // 256} tiles): streaming {t_old:.3f} ms, LDS-DMA incl. pre-split {t_new:.3f} ms  -> {t_old / t_new:.2f}x")
