"""Stand-alone ErwinFlash (ShapeNetCar configuration, benchmarks/02.1-Car-Design-ShapeNetCar/main.py:57-71) on one
synthetic cloud: ms per stage and per forward on the GPU, and the CPU oracle's forward beside it."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haet_amd import pc_ops
from haet_amd.components.erwinflash import ErwinTransformer, pointcloud

import gc  # (a full collection of the import-time heap is a ~75 ms stall inside a timed loop: tools/stall_probe.py)

gc.collect()
gc.freeze()

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=32186)
ap.add_argument("--radius", type=float, default=0.04)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--cpu", action="store_true", help="also time the CPU oracle (one forward)")
a = ap.parse_args()
dev = torch.device("cuda:0")
kw = dict(c_in=7, c_hidden=[128, 128], ball_sizes=[64, 32], enc_num_heads=[8, 8], enc_depths=[4, 4], dec_num_heads=[8], dec_depths=[4],
          strides=[2], rotate=45, decode=True, mlp_ratio=4, dimensionality=3, mp_steps=3)
torch.manual_seed(0)
m = ErwinTransformer(**kw).to(dev).eval()
g = torch.Generator().manual_seed(1)
N = a.points
pos_c, feat_c = torch.rand(N, 3, generator=g), torch.randn(N, 7, generator=g)
pos, feat = pos_c.to(dev), feat_c.to(dev)
bidx = torch.zeros(N, dtype=torch.long, device=dev)

def timed(fn, steps=a.steps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3, r

res = {"points": N, "radius": a.radius}
with torch.no_grad():
    res["tree_ms"], (ti, tm, rots) = timed(lambda: pc_ops.balltree_big(pos, [N], kw["strides"], kw["ball_sizes"], 45.0))
    res["radius_graph_ms"], ei = timed(lambda: pc_ops.radius_graph(pos, a.radius, bidx, [N]))
    res["edges"] = int(ei.shape[1])
    W = pointcloud._weights(m)
    res["embed_mpnn_ms"], _ = timed(lambda: pointcloud._embed(m, W, feat, pos, ei))
    res["forward_ms"], out = timed(lambda: m(feat, pos, bidx, radius=a.radius))
    res["forward_given_edges_ms"], _ = timed(lambda: m(feat, pos, bidx, edge_index=ei))
res["points_per_s"] = N / res["forward_ms"] * 1e3
if a.cpu:
    from oracle import haet_oracle as O
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    cfg = O.ErwinCfg(**{k: kw[k] for k in ("c_in", "c_hidden", "ball_sizes", "enc_num_heads", "enc_depths", "dec_num_heads", "dec_depths",
                                              "strides", "rotate", "decode", "mlp_ratio", "dimensionality")})
    t = time.perf_counter()
    O.build_balltree_with_rotations(pos_c, bidx.cpu(), kw["strides"], kw["ball_sizes"], 45.0)
    res["cpu_oracle_tree_ms"] = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    O.radius_graph(pos_c, a.radius, bidx.cpu())
    res["cpu_oracle_radius_graph_ms"] = (time.perf_counter() - t) * 1e3
    ei_c = ei.cpu()
    t = time.perf_counter()
    with torch.no_grad():
        ref = O.erwinflash_forward(feat_c, pos_c, bidx.cpu(), sd, cfg, 3, edge_index=ei_c)
    res["cpu_oracle_forward_given_edges_s"] = time.perf_counter() - t
    res["cpu_threads"] = torch.get_num_threads()
    res["rel_l2_vs_oracle"] = O.rel_l2(out.cpu(), ref)
print(json.dumps(res))
