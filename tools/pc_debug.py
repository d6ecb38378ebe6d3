"""Diagnostics of the stand-alone ErwinFlash path on the GPU box (not a test)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch, math
from oracle import haet_oracle as O
from oracle import balltree as obt
from tests import golden_util as G
from haet_amd import pc_ops
from haet_amd.components.erwinflash import ErwinTransformer, pointcloud
dev = torch.device("cuda:0")

for name in G.ef_fixture_names():
    f = G.load_ef(name); meta = f["meta"]
    m = ErwinTransformer(**meta["kw"]); m.load_state_dict(f["sd"]); m = m.to(dev).train(meta["train"])
    args = dict(edge_index=f["edge_index"].to(dev)) if meta["edges"] == "given" else dict(radius=meta["radius"])
    with torch.no_grad():
        out = m(f["feat"].to(dev), f["pos"].to(dev), f["batch_idx"].to(dev), **args)
    m2 = ErwinTransformer(**meta["kw"]); m2.load_state_dict(f["sd"]); m2 = m2.to(dev).train(meta["train"])
    with torch.no_grad():
        ref = pointcloud.forward_autograd(m2, f["feat"].to(dev), f["pos"].to(dev), f["batch_idx"].to(dev), args.get("edge_index"), meta["radius"])
    print(name, "hip vs reference %.2e" % O.rel_l2(out.cpu(), f["exp"]["out"]), " torch-device vs reference %.2e" % O.rel_l2(ref.cpu(), f["exp"]["out"]),
          " hip vs torch-device %.2e" % O.rel_l2(out.cpu(), ref.cpu()))

def analyse(sizes, d, kind):
    rng = np.random.default_rng(sum(sizes) + d); n = sum(sizes)
    if kind == "rand": p = rng.random((n, d))
    elif kind == "ties": p = rng.integers(0, 50, (n, d)).astype(np.float64)
    elif kind == "grid": p = np.round(rng.random((n, d)) * 64) / 64
    pos = torch.from_numpy(p.astype(np.float32))
    bidx = torch.cat([torch.full((s,), i, dtype=torch.long) for i, s in enumerate(sizes)])
    strides, balls = [2, 2], [256, 128, 64]
    eti, etm, erots = O.build_balltree_with_rotations(pos, bidx, strides, balls, 45.0)
    ti, tm, rots = pc_ops.balltree_big(pos.to(dev), sizes, strides, balls, 45.0)
    print(kind, sizes, "tree", torch.equal(ti.cpu().long(), eti))
    leaves = pos[eti]
    R = O.rotation_matrix(45.0, d)
    rot_t = torch.matmul(leaves, R)
    # fma-chain emulation of the device kernel
    L64, R64 = leaves.double(), R.double()
    acc = (L64[:, 0:1] * R64[0:1, :]).float()
    for k in range(1, d):
        acc = (L64[:, k:k+1] * R64[k:k+1, :] + acc.double()).float()
    print("  rotated leaves: torch.matmul vs fma chain differ in", int((rot_t != acc).sum()), "of", rot_t.numel())
    cur_t, cur_f = rot_t, acc
    for i, r in enumerate(rots):
        a, b = r.cpu().long(), erots[i]
        eq = torch.equal(a, b)
        nlev = a.numel()
        msg = ""
        if not eq:
            for g in (1 << s for s in range(1, 20)):
                if nlev % g: break
                sa = torch.sort(a.view(-1, g), dim=1)[0]; sb = torch.sort(b.view(-1, g), dim=1)[0]
                bad = int((sa != sb).any(dim=1).sum())
                msg += f" g{g}:{bad}"
        # the oracle partition on the fma-chain keys
        tgt = max(0, int(math.log2(eti.numel() / balls[i])))
        cb = bidx[eti]
        for s in strides[:i]: cb = cb[::s]
        alt = torch.from_numpy(obt.partition_balltree(cur_f.double().numpy(), cb.numpy(), tgt))
        print(f"  rot{i}: equal to oracle(torch.matmul keys) {eq}; equal to oracle(fma keys) {torch.equal(a, alt)}; mismatching groups by size:{msg}")
        if i < len(rots) - 1:
            cur_t = cur_t.view(-1, strides[i], d).mean(dim=1); cur_f = cur_f.view(-1, strides[i], d).mean(dim=1)

analyse([262144], 3, "rand")
analyse([40000], 3, "ties")
analyse([9000, 9000], 2, "grid")
