#!/usr/bin/env python3
"""Training step (forward + backward) of Physics_Attention_Structured_Mesh_3D at C5's widths on a G^3 grid:
HIP convolution kernels in all three directions against MIOpen (HAET_NO_HIP_CONV=1)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from haet_amd.PhysicsAttention import Physics_Attention_Structured_Mesh_3D

G = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dev = torch.device("cuda", 0)
torch.manual_seed(0)
mod = Physics_Attention_Structured_Mesh_3D(dim=256, heads=8, dim_head=32, slice_num=128, H=G, W=G, D=G, rotate=45).to(dev).train()
x = torch.randn(1, G ** 3, 256, device=dev, requires_grad=True)
gout = torch.randn(1, G ** 3, 256, device=dev)


def step():
    mod.zero_grad(set_to_none=True)
    x.grad = None
    mod(x).backward(gout)


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
mod.stage_events = []
step()
torch.cuda.synchronize()
st = {}
for n, a, b in mod.stage_events:
    st[n] = st.get(n, 0) + a.elapsed_time(b)
print(f"{'MIOpen' if os.environ.get('HAET_NO_HIP_CONV') else 'HIP'} conv, {G}^3 = {G**3} points: {1e3 * dt:.1f} ms / training step; "
      + ", ".join(f"{k} {v:.1f}" for k, v in st.items()) + f"; peak {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")
