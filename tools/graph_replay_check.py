#!/usr/bin/env python3
"""HIP-graph replay (torch.compile mode="reduce-overhead") of several model configurations against eager: which
configuration breaks on replay."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import haet_amd

dev = torch.device("cuda", 0)
I, S2, S3 = haet_amd.HAETransolver_Irregular_Mesh, haet_amd.HAETransolver_Structured_Mesh_2D, haet_amd.HAETransolver_Structured_Mesh_3D
CASES = [
    ("irregular dh16 M64", I, dict(space_dim=2, n_layers=2, n_hidden=128, n_head=8, fun_dim=0, out_dim=1, slice_num=64), 1891),
    ("irregular dh32 M64", I, dict(space_dim=2, n_layers=2, n_hidden=256, n_head=8, fun_dim=0, out_dim=1, slice_num=64), 1891),
    ("structured dh32 M64", S2, dict(space_dim=2, n_layers=2, n_hidden=256, n_head=8, fun_dim=0, out_dim=1, slice_num=64, H=61, W=31), 1891),
    ("structured dh16 M8", S2, dict(space_dim=2, n_layers=2, n_hidden=128, n_head=8, fun_dim=0, out_dim=1, slice_num=8, H=61, W=31), 1891),
    ("structured dh16 M64", S2, dict(space_dim=2, n_layers=2, n_hidden=128, n_head=8, fun_dim=0, out_dim=1, slice_num=64, H=61, W=31), 1891),
    ("irregular dh16 M128 (LDS tree builder)", I, dict(space_dim=2, n_layers=2, n_hidden=128, n_head=8, fun_dim=0, out_dim=1, slice_num=128), 972),
    ("C1: structured 64 x 64, n_hidden 256, M8", S2, dict(space_dim=2, n_layers=2, n_hidden=256, n_head=8, fun_dim=0, out_dim=1, slice_num=8, H=64, W=64), 4096),
    ("structured 3-D 12^3, n_hidden 128, M32", S3, dict(space_dim=3, n_layers=2, n_hidden=128, n_head=8, fun_dim=0, out_dim=1, slice_num=32, H=12, W=12, D=12), 1728),
    ("structured dh16 M64, 1 layer", S2, dict(space_dim=2, n_layers=1, n_hidden=128, n_head=8, fun_dim=0, out_dim=1, slice_num=64, H=61, W=31), 1891),
]
only = os.environ.get("ONLY")
for name, cls, kw, N in CASES:
    if only and only not in name:
        continue
    torch.manual_seed(0)
    torch._dynamo.reset()
    m = cls(**kw).eval().to(dev)
    x = torch.rand(2, N, kw["space_dim"], device=dev)
    with torch.no_grad():
        y0 = m(x, None)
        mc = torch.compile(m, mode="reduce-overhead")
        res = []
        for i in range(5):
            y1 = mc(x, None).clone()
            res.append(float((y1 - y0).norm() / y0.norm()))
    print(f"{name}: rel diff per call {['%.1e' % r for r in res]}", flush=True)
