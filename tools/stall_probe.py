#!/usr/bin/env python3
"""Where sporadic ~80 ms outliers in timed loops come from: per-iteration wall times of the attention module beside the
garbage collector's events -- a generation-2 collection walks the whole import-time heap (torch, sympy, ...) and takes ~75 ms
whatever it frees.  bench.py and the probes call gc.freeze() after set-up for that reason."""
import gc, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh as PA
ev = []
t_gc = [0.0]
def cb(phase, info):
    if phase == "start": t_gc[0] = time.perf_counter()
    else: ev.append((info["generation"], info["collected"], (time.perf_counter() - t_gc[0]) * 1e3))
gc.callbacks.append(cb)
dev = torch.device("cuda", 0)
for (H, C, M, N) in [(8, 32, 32, 100000), (8, 16, 64, 20000)]:
    torch.manual_seed(0)
    m = PA(dim=H * C, heads=H, dim_head=C, slice_num=M, rotate=45).eval().to(dev)
    x = torch.randn(1, N, H * C, device=dev)
    ts = []
    with torch.no_grad():
        for i in range(300):
            t0 = time.perf_counter()
            m(x)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
    big = [(i, round(t, 2)) for i, t in enumerate(ts) if t > 3 * sorted(ts)[len(ts) // 2]]
    print(H, C, M, N, "median %.3f ms" % sorted(ts)[len(ts) // 2], "outliers:", big[:10])
    print("gc events (gen, collected, ms):", [e for e in ev if e[2] > 1.0][:10], "n gc", len(ev))
    ev.clear()
