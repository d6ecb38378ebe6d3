#!/usr/bin/env python3
"""What do the per-stage timing events cost?  (host launch path with / without them)"""
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
steps = 50


def run(label, pool):
    with torch.no_grad():
        for _ in range(3):
            mod(x)
        mod.stage_events = [] if pool is not None else None
        mod.event_pool = pool if pool else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            mod(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ev = mod.stage_events
        mod.stage_events = None
        mod.event_pool = None
    msg = f"{label}: launch path {1e3*(t1-t0)/steps:.3f} ms, drained {1e3*(t2-t0)/steps:.3f} ms per forward"
    if ev:
        st = {}
        for name, e0, e1 in ev:
            st[name] = st.get(name, 0.0) + e0.elapsed_time(e1) / steps
        msg += "  " + str({k: round(v, 3) for k, v in st.items()})
    print(msg)


def run_like_bench(label):
    import bench as B
    with torch.no_grad():
        for _ in range(3):
            out = mod(x)
        mod.stage_events = []
        mod.event_pool = [torch.cuda.Event(enable_timing=True) for _ in range(14 * steps)]
        for e in mod.event_pool:
            e.record()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = mod(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        mod.stage_events = None
        mod.event_pool = None
    print(f"{label}: launch path {1e3*(t1-t0)/steps:.3f} ms, drained {1e3*(t2-t0)/steps:.3f} ms per forward")


if os.environ.get("LIKE_BENCH"):
    run_like_bench("like bench, first thing in the process")
run("no events", None)
run("fresh events (created lazily at first record)", [])
pool = [torch.cuda.Event(enable_timing=True) for _ in range(14 * steps)]
for e in pool:
    e.record()  # materialise the hipEvent
torch.cuda.synchronize()
run("pooled, pre-recorded events", pool)
run_like_bench("like bench, at the end")
