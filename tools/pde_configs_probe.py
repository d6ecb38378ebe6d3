#!/usr/bin/env python3
"""The seven configurations the reference's PDE-benchmark scripts launch
(benchmarks/04-PDE-Solving-StandardBenchmark/scripts/*.sh with the constructor calls of exp_{airfoil,darcy,elas,ns,pipe,plas}.py):
model class, widths, grid / mesh size and batch size as there, synthetic inputs.  Per configuration: eager eval forward and
one training step as the scripts run it (forward, rel-L2 loss, backward, gradient clipping, AdamW).  A drop-in check on
the users' real shapes (a latency regime: 1k-17k points, n_hidden 128) -- failures and cliffs show here first."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import haet_amd

dev = torch.device("cuda", 0)
S2, IRR = haet_amd.HAETransolver_Structured_Mesh_2D, haet_amd.HAETransolver_Irregular_Mesh
CONFIGS = [
    # name, class, kwargs, batch, points, fun_dim, T input, max_grad_norm
    ("airfoil M=64 (HAETransolver_Airfoil.sh)", S2, dict(space_dim=2, n_layers=8, n_hidden=128, n_head=8, fun_dim=0, out_dim=1, slice_num=64, ref=8, unified_pos=0, H=221, W=51), 4, 221 * 51, 0, False, 0.1),
    ("airfoil M=128 (HAET_Airfoil.sh)", S2, dict(space_dim=2, n_layers=8, n_hidden=128, n_head=8, fun_dim=0, out_dim=1, slice_num=128, ref=8, unified_pos=0, H=221, W=51), 4, 221 * 51, 0, False, 0.1),
    ("darcy 85x85 M=64 (HAET_Darcy.sh)", S2, dict(space_dim=2, n_layers=8, n_hidden=128, n_head=8, fun_dim=1, out_dim=1, slice_num=64, ref=8, unified_pos=1, H=85, W=85), 4, 85 * 85, 1, False, 0.1),
    ("elasticity 972 pts M=128 (HAETransolver_Elas.sh)", IRR, dict(space_dim=2, n_layers=8, n_hidden=128, n_head=8, fun_dim=0, out_dim=1, slice_num=128, ref=8, unified_pos=0), 1, 972, 0, False, 0.1),
    ("navier-stokes 64x64 M=32 (HAETransolver_NS.sh)", S2, dict(space_dim=2, n_layers=8, n_hidden=256, n_head=8, fun_dim=10, out_dim=1, slice_num=32, ref=8, unified_pos=1, H=64, W=64), 2, 64 * 64, 10, False, None),
    ("pipe 129x129 M=64 (HAETransolver_Pipe.sh)", S2, dict(space_dim=2, n_layers=8, n_hidden=128, n_head=8, mlp_ratio=2, fun_dim=0, out_dim=1, slice_num=64, ref=8, unified_pos=0, H=129, W=129), 8, 129 * 129, 0, False, 0.1),
    ("plasticity 101x31 M=64, Time_Input (HAETransolver_Plas.sh)", S2, dict(space_dim=2, n_layers=8, n_hidden=128, n_head=8, Time_Input=True, fun_dim=1, out_dim=4, slice_num=64, unified_pos=0, H=101, W=31), 8, 101 * 31, 1, True, 0.1),
]
only = os.environ.get("PDE_ONLY")
import gc

gc.collect()
gc.freeze()  # (a full collection of the import-time heap is a ~75 ms stall: tools/stall_probe.py)
for name, cls, kw, B, N, fd, tin, clip in CONFIGS:
    if only and only not in name:
        continue
    torch.manual_seed(0)
    model = cls(**kw).to(dev)
    x = torch.rand(B, N, 2, device=dev)
    fx = torch.rand(B, N, fd, device=dev) if fd else None
    T = torch.rand(B, 1, device=dev) if tin else None
    y = torch.randn(B, N, kw["out_dim"], device=dev)
    args = (x, fx, T) if tin else (x, fx)

    def fwd():
        return model(*args)

    model.eval()
    with torch.no_grad():
        for _ in range(3):
            out = fwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = fwd()
        torch.cuda.synchronize()
        t_eval = (time.perf_counter() - t0) / 10
    assert out.shape == (B, N, kw["out_dim"]) and bool(torch.isfinite(out).all()), name
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-5)

    def step():
        o = fwd()
        loss = ((o - y).reshape(B, -1).norm(dim=1) / y.reshape(B, -1).norm(dim=1)).mean()
        opt.zero_grad()
        loss.backward()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        opt.step()
        return loss

    for _ in range(4):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        loss = step()
    torch.cuda.synchronize()
    t_train = (time.perf_counter() - t0) / 20
    assert bool(torch.isfinite(loss)), name
    print(f"{name}: batch {B} x {N} points: eval forward {1e3 * t_eval:.2f} ms, training step {1e3 * t_train:.2f} ms "
          f"(loss {float(loss):.4f}, peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB)", flush=True)
    del model, opt
    torch.cuda.reset_peak_memory_stats()
