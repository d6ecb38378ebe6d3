#!/usr/bin/env python3
"""Race / order screen of the LDS tree builder (csrc/balltree.hip: round 4 -- eight waves per cloud, nodes of more than
four items selected by a whole wave): many random clouds of 2..128 tokens (uniform, tie-heavy, collinear) in ONE process,
the main tree and -- on tie-free clouds -- the rotated partitions compared with the CPU oracle.  GPU box only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["HAET_TREE_LDS"] = "1"  # (the generic builder for every size, power-of-two clouds included)
import numpy as np
import torch

from haet_amd import ops
from oracle import haet_oracle as O

dev = torch.device("cuda:0")
rng = np.random.default_rng(2026)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
bad = bad_rot = 0
t0 = time.time()
for case in range(n_cases):
    M = int(rng.integers(2, 129)) if case % 3 else int(rng.choice([64, 96, 100, 127, 128]))
    nc = int(rng.integers(1, 33))
    d = int(rng.integers(2, 4))
    kind = case % 4
    if kind == 0 or kind == 3:
        p = rng.random((nc * M, d))
    elif kind == 1:
        p = rng.integers(0, 6, (nc * M, d)).astype(np.float64)  # exact ties everywhere
    else:
        p = rng.random((nc * M, d))
        p[:, 1:] = 0.5  # collinear
    pos = torch.from_numpy(p.astype(np.float32))
    bidx = torch.arange(nc).repeat_interleave(M)
    b0 = 1 << (min(32, M).bit_length() - 1)  # powers of two: the level sizes are multiples of them
    balls = [b0, max(1, b0 // 2)]
    angle = 45.0 if kind == 3 else 0.0
    eti, etm, erots = O.build_balltree_with_rotations(pos, bidx, [2], balls, angle)
    ti, tm, rot, sizes = ops.balltree_build(pos.to(dev), nc, M, [2], balls, angle)
    ok = torch.equal(ti.cpu().long(), eti) and torch.equal(tm.cpu().bool(), etm)
    bad += not ok
    if angle > 0 and ok:  # (the rotated keys are fp32 products: balls can differ where two keys are an ulp apart)
        off = 0
        for i, n in enumerate(sizes):
            g = balls[i]
            if n % g:
                off += n
                continue
            a = torch.sort(rot[off:off + n].cpu().long().view(-1, g), dim=1)[0]
            b = torch.sort(erots[i].view(-1, g), dim=1)[0]
            bad_rot += int((a != b).any())
            off += n
print(f"{n_cases} cases, {bad} main-tree mismatches, {bad_rot} rotated levels with a differing ball, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
