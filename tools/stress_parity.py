#!/usr/bin/env python3
"""Randomised parity sweep of the slice / de-slice passes and their backward against fp64 torch on random shapes
(all kernel families pick themselves by shape).  Prints the worst relative errors; exits non-zero on a failure."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from haet_amd import ops

dev = torch.device("cuda", 0)
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
gen = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))


worst = {}
fails = 0
for it in range(n_cfg):
    C = [8, 16, 32, 32, 32, 64, 64][ri(0, 6)]
    M = [8, 16, 32, 32, 48, 64, 64, 100, 128][ri(0, 8)]
    H = [1, 2, 3, 4, 8, 8, 16][ri(0, 6)]
    B = ri(1, 3)
    N = [1, 31, 32, 33, 255, 256, 257, 1000, 4095, 4096, 4097, 20000][ri(0, 11)] if ri(0, 1) else ri(1, 30000)
    x = torch.randn(B, N, H, C, generator=gen)
    Ws = torch.randn(M, C, generator=gen) * (1.6 / C ** 0.5)
    bs = torch.randn(M, generator=gen) * 0.3
    wt = torch.randn(C, generator=gen) * (1.2 / C ** 0.5)
    bt = 0.05
    S2 = torch.randn(B, H, M, C, generator=gen)
    go = torch.randn(B, N, H, C, generator=gen)
    dS = torch.randn(B, H, M, C, generator=gen)
    sp = ops.SliceParams(Ws.to(dev), bs.to(dev), wt.to(dev), bt, 0.5, 1e-6)
    L = [t.double().requires_grad_(True) for t in (x, Ws, bs, wt, torch.tensor(bt), S2)]
    xp = L[0].permute(0, 2, 1, 3)
    tau = 0.5 + torch.clamp(xp @ L[3] + L[4], -0.4, 0.4)
    z = (xp @ L[1].t() + L[2] - sp.c0) / tau[..., None]
    w = torch.softmax(z, dim=2)
    S = torch.einsum("bhnc,bhng->bhgc", xp, w) / (w.sum(2) + 1e-5)[..., None]
    out = torch.einsum("bhgc,bhng->bhnc", L[5], w).permute(0, 2, 1, 3)
    loss = (out * go.double()).sum() + (S * dS.double()).sum()
    gx, gWs, gbs, gwt, gbt, gS2 = torch.autograd.grad(loss, L)
    x_lin = x.reshape(B, N, H * C).to(dev)
    cm, cs, S_hip = ops.merge_records(ops.slice_partials(x_lin, sp, H, C, M), B, H, C, M, True)
    out_hip = ops.deslice(x_lin, sp, cm, cs, S2.reshape(-1, C).to(dev), H, C, M)
    god = go.reshape(B, N, H * C).to(dev)
    dS2 = ops.deslice_bwd_tokens(x_lin, sp, cm, cs, god, H, C, M)
    S2d, dSd = S2.reshape(-1, C).to(dev), dS.reshape(-1, C).to(dev)
    A = (dS2 * S2d).sum(-1) + (S_hip * dSd).sum(-1)
    dx, dWs, dbs, dwt, dbt = ops.slice_bwd_points(x_lin, sp, cm, cs, god, S2d, dSd, A, H, C, M)

    def rel(a, b):
        return float((a.cpu().double().reshape(b.shape) - b).norm() / b.norm().clamp_min(1e-30))

    errs = {"S": rel(S_hip, S.detach()), "out": rel(out_hip, out.detach()), "dS2": rel(dS2, gS2), "dx": rel(dx, gx)}
    if N > 1:
        errs["dWs"] = rel(dWs, gWs)
    bad = {k: v for k, v in errs.items() if not v < 2e-3}
    if bad:
        fails += 1
        print(f"FAIL B={B} N={N} H={H} C={C} M={M}: {bad}", flush=True)
    for k, v in errs.items():
        if v > worst.get(k, (0, None))[0]:
            worst[k] = (v, (B, N, H, C, M))
print("worst:", {k: (f"{v[0]:.2e}", v[1]) for k, v in worst.items()}, "configs", n_cfg, "failures", fails)
sys.exit(1 if fails else 0)
