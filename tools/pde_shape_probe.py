#!/usr/bin/env python3
"""The reference's own PDE-benchmark widths (benchmarks/04-PDE-Solving-StandardBenchmark/scripts/*.sh: n_hidden 128, 8 heads ->
dim_head 16, 8 layers, slice_num 32 / 64 / 128) at their mesh sizes (~1k .. 17k points): model forward and the attention
module's stage times -- a latency regime, not a bandwidth one."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from haet_amd import HAETransolver_Irregular_Mesh
from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh

dev = torch.device("cuda", 0)
import gc  # (a full collection of the import-time heap is a ~75 ms stall inside a timed loop: tools/stall_probe.py)

gc.collect()
gc.freeze()
for N, M in [(972, 128), (11271, 64), (16641, 64)]:
    torch.manual_seed(0)
    model = HAETransolver_Irregular_Mesh(space_dim=2, fun_dim=0, n_hidden=128, n_layers=8, n_head=8, mlp_ratio=1, out_dim=1,
                                         slice_num=M).eval().to(dev)
    x = torch.rand(1, N, 2, device=dev)
    with torch.no_grad():
        for _ in range(3):
            model(x, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model(x, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    mod = Physics_Attention_Irregular_Mesh(dim=128, heads=8, dim_head=16, slice_num=M, rotate=45).eval().to(dev)
    xa = torch.randn(1, N, 128, device=dev)
    with torch.no_grad():
        for _ in range(3):
            mod(xa)
        mod.stage_events = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            mod(xa)
        torch.cuda.synchronize()
        da = (time.perf_counter() - t0) / 10
    st = {}
    for n, a, b in mod.stage_events:
        st[n] = st.get(n, 0) + a.elapsed_time(b) / 10
    print(f"N={N} M={M}: 8-layer model forward {1e3 * dt:.3f} ms; attention module {1e3 * da:.3f} ms: " + ", ".join(f"{k} {v:.3f}" for k, v in st.items()), flush=True)
