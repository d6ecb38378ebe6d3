"""The uniform-grid radius search against the all-pairs kernel (same selection rule, independent candidate enumeration)
on clouds the CPU oracle cannot reach: random sizes / radii / dimensions / offsets, several clouds.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from haet_amd import pc_ops

dev = torch.device("cuda:0")
rng = np.random.default_rng(777)
bad = 0
t0 = time.time()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for rep in range(reps):
    d = int(rng.integers(2, 4))
    nc = int(rng.integers(1, 4))
    sizes = [int(rng.integers(2, 60000)) for _ in range(nc)]
    n = sum(sizes)
    scale = float(10.0 ** rng.uniform(-2, 2))
    offset = float(rng.uniform(-5, 5)) * scale
    p = (rng.random((n, d)) * scale + offset).astype(np.float32)
    if rep % 3 == 1:
        p = np.round(p / scale * 64).astype(np.float32) * np.float32(scale / 64)  # many exactly equal distances
    pos = torch.from_numpy(p).to(dev)
    bidx = torch.cat([torch.full((s,), i, dtype=torch.long) for i, s in enumerate(sizes)]).to(dev)
    dens = n / nc / scale ** d
    r = float((rng.uniform(2, 60) / dens) ** (1.0 / d))
    a = pc_ops.radius_graph(pos, r, bidx, sizes, grid=True)
    b = pc_ops.radius_graph(pos, r, bidx, sizes, grid=False)
    ok = torch.equal(a, b)
    if not ok:
        bad += 1
        print("MISMATCH", rep, sizes, d, r, a.shape, b.shape, flush=True)
print(f"{reps} clouds, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
