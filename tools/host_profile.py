#!/usr/bin/env python3
"""Host-side cost of one forward: cProfile over many small forwards (the GPU work of a 32k-point
cloud is ~0.4 ms, anything above that per call is the Python/ctypes launch path)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda:0")
mod = bench.build_module(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
x = torch.randn(1, n, 256, device=dev)
with torch.no_grad():
    for _ in range(5):
        mod(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        mod(x)
    t1 = time.perf_counter()  # launch path only: no sync inside the loop
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={n}: host launch path {1e3*(t1-t0)/200:.3f} ms/forward, with the GPU drained {1e3*(t2-t0)/200:.3f} ms/forward")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        mod(x)
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
