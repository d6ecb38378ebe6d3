"""Race screen of the big-cloud tree builder (its rounds hand data between waves through a few barriers): many random
clouds of several sizes / kinds in ONE process, every permutation compared with the CPU oracle.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import balltree as obt
from haet_amd import pc_ops

dev = torch.device("cuda:0")
rng = np.random.default_rng(12345)
bad = 0
t0 = time.time()
cases = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    d = int(rng.integers(2, 4))
    nc = int(rng.integers(1, 4))
    sizes = [int(rng.integers(2, 200000 if rep % 5 else 600000)) for _ in range(nc)]
    n = sum(sizes)
    kind = rep % 4
    if kind == 0:
        p = rng.random((n, d))
    elif kind == 1:
        p = rng.integers(0, 40, (n, d)).astype(np.float64)
    elif kind == 2:
        p = np.round(rng.random((n, d)) * 128) / 128
    else:
        p = rng.random((n, d)); p[:, 0] = 0.5
    pos = torch.from_numpy(p.astype(np.float32))
    bidx = torch.cat([torch.full((s,), i, dtype=torch.long) for i, s in enumerate(sizes)])
    ti, tm = obt.build_balltree(pos.double().numpy(), bidx.numpy())
    gi, gm, _ = pc_ops.balltree_big(pos.to(dev), sizes, [], [64], 0.0)
    ok = np.array_equal(gi.cpu().numpy().astype(np.int64), ti) and np.array_equal(gm.cpu().numpy().astype(bool), tm)
    cases += 1
    if not ok:
        bad += 1
        print("MISMATCH", rep, sizes, d, kind, flush=True)
print(f"{cases} clouds, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
