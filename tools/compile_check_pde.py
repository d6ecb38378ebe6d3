#!/usr/bin/env python3
"""torch.compile over a PDE-width model (dim_head 16: the padded matrix-core passes inside the opaque operators): compiled
inference (default mode and reduce-overhead = HIP graphs) against eager, and a compiled training step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import haet_amd

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = haet_amd.HAETransolver_Structured_Mesh_2D(space_dim=2, n_layers=2, n_hidden=128, n_head=8, fun_dim=0, out_dim=1, slice_num=64,
                                              ref=8, unified_pos=0, H=61, W=31).eval().to(dev)
x = torch.rand(2, 61 * 31, 2, device=dev)
with torch.no_grad():
    y0 = m(x, None)
    print("padded:", m.blocks[0].Attn.head_pad_applies(2, 61 * 31), "eager finite:", bool(torch.isfinite(y0).all()), flush=True)
    for mode in (None, "reduce-overhead"):
        torch._dynamo.reset()
        mc = torch.compile(m, mode=mode) if mode else torch.compile(m)
        for i in range(4):
            y1 = mc(x, None)
            print(f"mode={mode} call {i}: finite {bool(torch.isfinite(y1).all())}, rel diff vs eager {float((y1 - y0).norm() / y0.norm()):.2e}", flush=True)
m.train()
torch._dynamo.reset()
mt = torch.compile(m)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
for _ in range(3):
    loss = mt(x, None).square().mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
print("compiled training step ok, loss", float(loss.detach()))
