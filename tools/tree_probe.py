#!/usr/bin/env python3
"""Time of the LDS tree builder (csrc/balltree.hip: clouds of 65..128 tokens and the non-power-of-two ones) alone: M = 128
tokens per cloud, the PDE benchmark's 8 clouds and C5's 8 / 64, and a non-power-of-two size.  HAET_TREE_LANE_NODE (node size
up to which a lane selects alone; read once per process) is the sweep knob."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from haet_amd import ops

dev = torch.device("cuda", 0)
for M, nc, d in [(128, 8, 3), (128, 64, 3), (128, 8, 2), (100, 8, 3), (48, 8, 3)]:
    torch.manual_seed(M + nc)
    pos = torch.rand(nc * M, d, device=dev)
    bs = [min(32, M), max(1, min(16, M // 2))]
    for _ in range(5):
        ops.balltree_build(pos, nc, M, [2], bs, 45.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        ops.balltree_build(pos, nc, M, [2], bs, 45.0)
    b.record()
    torch.cuda.synchronize()
    print(f"lane_node={os.environ.get('HAET_TREE_LANE_NODE', 'default')} M={M} clouds={nc} d={d}: {a.elapsed_time(b) / 50 * 1e3:.1f} us / build", flush=True)
