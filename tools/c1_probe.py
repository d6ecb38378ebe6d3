#!/usr/bin/env python3
"""BASELINE config C1: HAETransolver_Structured_Mesh_2D, 64 x 64 grid (N = 4096), M = 8 slices, one layer --
a latency case (T = 64 slice tokens); forward time of the model and of its attention module."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import haet_amd

import gc  # (a full collection of the import-time heap is a ~75 ms stall inside a timed loop: tools/stall_probe.py)

gc.collect()
gc.freeze()

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = haet_amd.HAETransolver_Structured_Mesh_2D(space_dim=2, fun_dim=1, n_layers=1, slice_num=8, H=64, W=64).eval().to(dev)
x, fx = torch.rand(1, 4096, 2, device=dev), torch.rand(1, 4096, 1, device=dev)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


with torch.no_grad():
    t_model = timed(lambda: model(x, fx))
    attn = model.blocks[0].Attn
    h = torch.randn(1, 4096, 256, device=dev)
    t_attn = timed(lambda: attn(h))
print(f"C1 (64 x 64, M = 8, 1 layer): model forward {t_model:.3f} ms ({4096 / t_model * 1e3:.3e} points/s), attention module {t_attn:.3f} ms")
with torch.no_grad():
    attn.stage_events = []
    for _ in range(10):
        attn(h)
    torch.cuda.synchronize()
    st = {}
    for n, a, b in attn.stage_events:
        st[n] = st.get(n, 0) + a.elapsed_time(b) / 10
    print("attention stages (ms): " + ", ".join(f"{k} {v:.3f}" for k, v in st.items()))
