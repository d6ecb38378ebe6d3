#!/usr/bin/env python3
"""Model-level forward (secondary workload): HAETransolver_Irregular_Mesh, one layer, 1M points.
Times the whole forward and its parts (HIP events) to show what the stock-torch callers around the path cost."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import haet_amd
from haet_amd import _fast


import gc  # (a full collection of the import-time heap is a ~75 ms stall inside a timed loop: tools/stall_probe.py)

gc.collect()
gc.freeze()

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--train", action="store_true", help="time a training step (forward + backward, train mode)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = haet_amd.HAETransolver_Irregular_Mesh(space_dim=3, fun_dim=4, n_hidden=256, n_layers=args.layers, n_head=8,
                                                  mlp_ratio=2, out_dim=4, slice_num=64).eval().to(dev)
    x = torch.randn(1, args.points, 7, device=dev)

    def timed(fn, *a):
        for _ in range(2):
            out = fn(*a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fn(*a)
        torch.cuda.synchronize()
        return out, 1e3 * (time.perf_counter() - t0) / args.steps

    if args.train:
        model.train()
        y = torch.randn(1, args.points, 4, device=dev)

        def step():
            model.zero_grad(set_to_none=True)
            torch.nn.functional.mse_loss(model(x, None), y).backward()

        _, t = timed(step)
        print(f"model training step ({args.layers} layer(s), {args.points} points): {t:.2f} ms, "
              f"peak memory {torch.cuda.max_memory_allocated() / 1e9:.1f} GB")
        return
    with torch.no_grad():
        _, t_all = timed(lambda: model(x, None))
        blk = model.blocks[0]
        fx, t_pre = timed(lambda: _fast.mlp(model.preprocess, x, model.placeholder))
        h, t_ln1 = timed(lambda t: _fast.layer_norm(blk.ln_1, t), fx)
        a, t_attn = timed(blk.Attn, h)
        fx2 = a + fx
        h2, t_ln2 = timed(lambda t: _fast.layer_norm(blk.ln_2, t), fx2)
        m, t_mlp = timed(blk.mlp, h2)
        _, t_res = timed(lambda: m + fx2)
        t_last = 0.0
        if blk.last_layer:
            _, t_last = timed(lambda: _fast.linear(blk.mlp2, _fast.layer_norm(blk.ln_3, fx2)))
    print(f"model forward {t_all:.2f} ms ({args.points / t_all * 1e3:.3e} points/s): preprocess {t_pre:.2f}, ln_1 {t_ln1:.2f}, "
          f"Attn {t_attn:.2f}, ln_2 {t_ln2:.2f}, mlp {t_mlp:.2f}, residual add {t_res:.2f} (x2), ln_3 + mlp2 {t_last:.2f}")


if __name__ == "__main__":
    main()
