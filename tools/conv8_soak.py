#!/usr/bin/env python3
"""Race / edge screen of the LDS-DMA convolution (csrc/conv8.hip): random grids, batch sizes, channel counts and kernel sizes,
each run three times -- the three outputs must be BIT-identical (an LDS-DMA piece read before it has landed, or overwritten
while a wave still reads it, shows as a tile that differs between runs) -- and equal to the streaming kernel of rounds 1-3
(the same arithmetic in the same summation order: also bit-identical, unless that kernel takes its split-k path) to 1e-5."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from haet_amd import ops

dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for i in range(cases):
    nd = random.choice([2, 3])
    k = random.choice([1, 3, 3, 3, 5]) if nd == 2 else random.choice([1, 3, 3])
    grid = tuple(random.randint(max(2, k // 2 + 1), 70 if nd == 2 else 24) for _ in range(nd))
    B = random.choice([1, 1, 2, 3])
    Cin = 32 * random.randint(1, 8)
    N = 1
    for g in grid:
        N *= g
    torch.manual_seed(i)
    x = torch.randn(B, N, Cin, device=dev) * random.choice([0.1, 1.0, 5.0])
    W = torch.randn(256, Cin, *([k] * nd), device=dev) * 0.05
    w = ops.SplitConv(W, torch.randn(256, device=dev))
    outs = [ops.conv_project_cells(x, w, grid) for _ in range(3)]
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    os.environ["HAET_CONV_V1"] = "1"
    os.environ["HAET_CONV_NO_SPLITK"] = "1"
    old = ops.conv_project(x, w, grid)
    del os.environ["HAET_CONV_V1"]
    err = float((outs[0] - old).norm() / old.norm())
    ok = same and err < 1e-5 and bool(torch.isfinite(outs[0]).all())
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} B={B} grid={grid} Cin={Cin} k={k}: rows {B * N}, three runs identical: {same}, vs streaming kernel {err:.1e}", flush=True)
print(f"{cases - bad} / {cases} cases clean")
sys.exit(1 if bad else 0)
