#!/usr/bin/env python3
"""Does capturing the forward in a HIP graph (torch.cuda.CUDAGraph) pay?  Small clouds, where the
launch gaps are the largest share."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
mod = bench.build_module(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
x = torch.randn(1, n, 256, device=dev)
steps = 100
with torch.no_grad():
    for _ in range(70):
        y_ref = mod(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        y_ref = mod(x)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps * 1e3
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            mod(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        y = mod(x)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / steps * 1e3
print(f"N={n}: eager {eager:.4f} ms, graph replay {graphed:.4f} ms per forward; outputs equal: {torch.equal(y, y_ref)}")
