#!/usr/bin/env python3
"""Micro-benchmark of the slice / de-slice passes at the C3 shape for a few chunk counts."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import ops

dev = torch.device("cuda:0")
B, N, H, C, M = 1, 1_000_000, 8, 32, 64
torch.manual_seed(0)
x = torch.randn(B, N, H * C, device=dev) * 0.3
sp = ops.SliceParams(torch.randn(M, C, device=dev) * 0.2, torch.zeros(M, device=dev),
                     torch.randn(1, C, device=dev) * 0.2, 0.0, 0.5, 1e-6)
S2 = torch.randn(B * H * M, C, device=dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for chunks in ((512,) if os.environ.get("PROBE_ONE") else (256, 512, 768, 1024, 2048)):
    os.environ["HAET_SLICE_CHUNKS"] = str(chunks)
    parts = ops.slice_partials(x, sp, H, C, M)
    cm, cs, S = ops.merge_records(parts, B, H, C, M, True)
    ts = timeit(lambda: ops.slice_partials(x, sp, H, C, M))
    td = timeit(lambda: ops.deslice(x, sp, cm, cs, S2, H, C, M))
    print(f"chunks={chunks:5d} (P={parts.shape[0]}): slice {ts:.3f} ms = {N*1024/ts/1e9:.2f} TB/s ; "
          f"deslice {td:.3f} ms = {N*2048/td/1e9:.2f} TB/s")
