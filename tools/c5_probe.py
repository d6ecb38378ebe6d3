#!/usr/bin/env python3
"""BASELINE config C5, one rank's share: Physics_Attention_Structured_Mesh_3D on a 32 x 256 x 256 slab
(2 097 152 points, dim 256, 8 heads, M = 128), forward stage times on one GPU (no halo: stand-alone slab)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from haet_amd.PhysicsAttention import Physics_Attention_Structured_Mesh_3D

import gc  # (a full collection of the import-time heap is a ~75 ms stall inside a timed loop: tools/stall_probe.py)

gc.collect()
gc.freeze()

dev = torch.device("cuda", 0)
planes = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
mod = Physics_Attention_Structured_Mesh_3D(dim=256, heads=8, dim_head=32, slice_num=128, H=planes, W=256, D=256,
                                           rotate=45).eval().to(dev)
N = planes * 256 * 256
x = torch.randn(1, N, 256, device=dev)
with torch.no_grad():
    for _ in range(2):
        mod(x)
    mod.stage_events = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = mod(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
st = {}
for n, a, b in mod.stage_events:
    st[n] = st.get(n, 0) + a.elapsed_time(b) / 5
print(f"C5 slab {planes}x256x256 = {N} points: {1e3 * dt:.2f} ms / forward ({N / dt:.3e} points/s), finite={bool(torch.isfinite(y).all())}; "
      + ", ".join(f"{k} {v:.3f}" for k, v in st.items()) + f"; peak {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")
