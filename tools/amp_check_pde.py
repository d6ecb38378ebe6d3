#!/usr/bin/env python3
"""The reference's scripts run under AMP (exp_ns.py:162 autocast + GradScaler; train.py:78): PDE-width models under
torch.autocast('cuda') -- eval forward against the fp32 forward, and training steps with a GradScaler (finite loss and
gradients, scaler not backing off)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import haet_amd

dev = torch.device("cuda", 0)
CASES = [
    ("structured 2-D 61 x 31, n_hidden 128, M 64", haet_amd.HAETransolver_Structured_Mesh_2D,
     dict(space_dim=2, n_layers=2, n_hidden=128, n_head=8, fun_dim=1, out_dim=1, slice_num=64, H=61, W=31), 61 * 31),
    ("irregular 972 points, n_hidden 128, M 128", haet_amd.HAETransolver_Irregular_Mesh,
     dict(space_dim=2, n_layers=2, n_hidden=128, n_head=8, fun_dim=1, out_dim=1, slice_num=128), 972),
    ("irregular 5000 points, n_hidden 256, M 64", haet_amd.HAETransolver_Irregular_Mesh,
     dict(space_dim=3, n_layers=2, n_hidden=256, n_head=8, fun_dim=1, out_dim=1, slice_num=64), 5000),
]
for name, cls, kw, N in CASES:
    torch.manual_seed(0)
    m = cls(**kw).to(dev)
    x, fx = torch.rand(2, N, kw["space_dim"], device=dev), torch.rand(2, N, 1, device=dev)
    y = torch.randn(2, N, 1, device=dev)
    m.eval()
    with torch.no_grad():
        ref = m(x, fx)
        with torch.autocast("cuda"):
            amp = m(x, fx)
    rel = float((amp.float() - ref).norm() / ref.norm())
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    scaler = torch.amp.GradScaler("cuda")
    s0 = scaler.get_scale()
    finite = []
    for _ in range(16):
        with torch.autocast("cuda"):
            loss = ((m(x, fx).float() - y) ** 2).mean()
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.1)
        scaler.step(opt)
        scaler.update()
        finite.append(bool(torch.isfinite(gn)))
    # (the scaler starts at 2^16 and backs off while the scaled fp16 gradients overflow: the first steps are skipped, as in the
    #  reference; what matters is that it settles)
    ok = bool(torch.isfinite(loss)) and all(finite[-6:])
    print(f"{name}: autocast eval vs fp32 rel diff {rel:.1e} (dtype {amp.dtype}); 16 AMP training steps: loss {float(loss.detach()):.4f}, "
          f"grad norm {float(gn):.3e}, scale {s0:.0f} -> {scaler.get_scale():.0f}, steps with finite gradients {sum(finite)}/16  {'OK' if ok and rel < 2e-2 else 'CHECK'}", flush=True)
