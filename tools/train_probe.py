#!/usr/bin/env python3
"""Time one training step (forward + backward) of Physics_Attention_Irregular_Mesh at the bench config.

    python tools/train_probe.py [--points 1000000] [--steps 10] [--eval] [--torch]

--torch runs the previous interim path (HAET_TORCH_BACKWARD=1: stock-torch restatement) for comparison;
--eval keeps the module in eval mode (gradients through frozen BatchNorm statistics).
Prints ms per step, peak memory and the step's HBM-floor estimate.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


import gc  # (a full collection of the import-time heap is a ~75 ms stall inside a timed loop: tools/stall_probe.py)

gc.collect()
gc.freeze()

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--torch", action="store_true")
    args = ap.parse_args()
    if args.torch:
        os.environ["HAET_TORCH_BACKWARD"] = "1"
    import torch

    from bench import build_module

    dev = torch.device("cuda", 0)
    mod = build_module(dev)
    mod.train(not args.eval)
    x = torch.randn(1, args.points, 256, device=dev, requires_grad=True)
    gout = torch.randn(1, args.points, 256, device=dev)

    def step():
        mod.zero_grad(set_to_none=True)
        x.grad = None
        y = mod(x)
        y.backward(gout)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # forward only, same mode, for the split
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            mod(x)
        torch.cuda.synchronize()
        dtf = (time.perf_counter() - t0) / args.steps
    # per-stage HIP events of a few more steps
    mod.stage_events = []
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    stage = {}
    for name, e0, e1 in mod.stage_events:
        stage[name] = stage.get(name, 0.0) + e0.elapsed_time(e1) / 3
    mod.stage_events = None
    print("stages (ms): " + ", ".join(f"{k} {v:.3f}" for k, v in stage.items()) + f"; sum {sum(stage.values()):.2f}")
    row = args.points * 256 * 4 / 1e9
    print(f"path={'torch' if args.torch else 'hip'} mode={'eval' if args.eval else 'train'} points={args.points}: "
          f"{1e3 * dt:.2f} ms / step (forward alone {1e3 * dtf:.2f} ms), {args.points / dt:.3e} points/s, "
          f"peak memory {torch.cuda.max_memory_allocated() / 1e9:.2f} GB ({torch.cuda.max_memory_allocated() / 1e9 / row:.1f} rows of N x 256 fp32)")


if __name__ == "__main__":
    main()
