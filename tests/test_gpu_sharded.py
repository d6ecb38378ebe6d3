"""GPU (-m gpu): the point-sharded module path end to end with TWO processes on one GPU
(gloo rendezvous; RCCL needs one device per rank, which the 8-GPU bench has).  Each rank owns
half of the points of a golden fixture, runs the HIP path with shard_group set and must
reproduce its half of the REFERENCE output."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from haet_amd.sharding import shard_bounds, shard_points
    from oracle import haet_oracle as O
    from tests import golden_util as G
    from tests.test_modules_cpu import CTOR

    dev = torch.device("cuda:0")
    errs = {}
    for name in ("irr_c3like_amp", "irr_small_amp", "irr_b2_amp"):
        x, sd, meta, exp, rows = G.load_attn(name)
        mod = CTOR[meta["kind"]](**meta["kw"])
        mod.load_state_dict(sd)
        mod = shard_points(mod.to(dev).eval())
        a, b = shard_bounds(x.shape[1], rank, world)
        with torch.no_grad():
            out = mod(x[:, a:b].contiguous().to(dev))
        errs[name] = O.rel_l2(out.cpu(), exp["out"][:, a:b])
    torch.save(errs, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_point_sharding_matches_reference(tmp_path, hip_device):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        for name, err in torch.load(os.path.join(tmp_path, f"r{r}.pt")).items():
            assert err < 1e-3, (r, name, err)
            assert err < 2e-4, (r, name, err)


def _structured_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from haet_amd.PhysicsAttention import Physics_Attention_Structured_Mesh_2D, Physics_Attention_Structured_Mesh_3D
    from haet_amd.sharding import shard_bounds, shard_points
    from oracle import haet_oracle as O

    dev = torch.device("cuda:0")
    errs = {}
    for kind, grid in (("structured2d", (12, 10)), ("structured3d", (6, 5, 4))):
        torch.manual_seed(7)
        if kind == "structured2d":
            mod = Physics_Attention_Structured_Mesh_2D(64, heads=4, dim_head=32, slice_num=32, H=grid[0], W=grid[1], rotate=45)
        else:
            mod = Physics_Attention_Structured_Mesh_3D(64, heads=4, dim_head=32, slice_num=32, H=grid[0], W=grid[1],
                                                       D=grid[2], rotate=45)
        mod = mod.eval()
        sd = {k: v.clone() for k, v in mod.state_dict().items()}
        N = 1
        for g in grid:
            N *= g
        plane = N // grid[0]
        x = torch.randn(2, N, 64)
        cfg = O.AttnCfg(dim=64, heads=4, dim_head=32, slice_num=32, kind=kind, grid=grid, rotate=45)
        ref = O.physics_attention_forward(x, sd, cfg)
        a, b = shard_bounds(grid[0], rank, world)        # whole planes of the first grid axis
        mod = shard_points(mod.to(dev))
        with torch.no_grad():
            out = mod(x[:, a * plane:b * plane].contiguous().to(dev))
        errs[kind] = O.rel_l2(out.cpu(), ref[:, a * plane:b * plane])
    torch.save(errs, os.path.join(tmp, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_slab_sharding_of_structured_meshes(tmp_path, hip_device):
    """Structured variants sharded by slabs of the first grid axis: halo planes exchanged between the
    ranks for the Conv2d / Conv3d projection, slice records all-gathered; each rank reproduces its
    slab of the unsharded CPU forward."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_structured_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        for name, err in torch.load(os.path.join(tmp_path, f"s{r}.pt")).items():
            assert err < 2e-4, (r, name, err)


def _grad_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from haet_amd import _autograd
    from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh
    from haet_amd.sharding import shard_bounds, shard_points

    dev = torch.device("cuda:0")
    out = {}
    for train in (False, True):
        torch.manual_seed(21)
        mod = Physics_Attention_Irregular_Mesh(dim=256, heads=8, dim_head=32, slice_num=64, rotate=45)
        with torch.no_grad():
            for name, p in mod.erwin.named_parameters():
                if p.dim() >= 2:
                    p.mul_(3.0)
        mod = mod.to(dev).train(train)
        x = torch.randn(1, 2500, 256).to(dev)
        gout = torch.randn(1, 2500, 256).to(dev)
        state = {k: v.clone() for k, v in mod.state_dict().items()}
        # reference: the whole cloud on this rank, autograd through the device restatement
        xr = x.clone().requires_grad_(True)
        (_autograd.attention_forward(mod, xr) * gout).sum().backward()
        ref = {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}
        ref_gx = xr.grad.clone()
        mod.zero_grad(set_to_none=True)
        mod.load_state_dict(state)
        # sharded: this rank's half of the points, the HIP autograd node with one all-gather + one all-reduce
        shard_points(mod)
        a, b = shard_bounds(x.shape[1], rank, world)
        xs = x[:, a:b].contiguous().requires_grad_(True)
        (mod(xs) * gout[:, a:b]).sum().backward()
        out[train] = dict(gx=xs.grad.cpu(), ref_gx=ref_gx[:, a:b].cpu(),
                          grads={n: p.grad.cpu() for n, p in mod.named_parameters() if p.grad is not None},
                          ref={n: g.cpu() for n, g in ref.items()})
    torch.save(out, os.path.join(tmp, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_point_sharded_gradients(tmp_path, hip_device):
    """Training on the point-sharded path: every rank back-propagates its own points (one all-reduce of
    dS2); its input gradient is its slice of the unsharded one and the ranks' parameter gradients are
    additive shares of the unsharded gradient (haet_amd/sharding.py)."""
    from oracle import haet_oracle as O

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(os.path.join(tmp_path, f"g{r}.pt")) for r in range(2)]
    for train in (False, True):
        for r in range(2):
            assert O.rel_l2(res[r][train]["gx"], res[r][train]["ref_gx"]) < 1e-3, (train, r)
        ref = res[0][train]["ref"]
        scale = max(float(g.norm()) for g in ref.values())
        for name, g_ref in ref.items():
            total = res[0][train]["grads"][name] + res[1][train]["grads"][name]
            diff = float((total - g_ref).norm())
            err = diff / max(float(g_ref.norm()), 1e-4 * scale)
            # (a bias in front of a train-mode BatchNorm has a gradient that cancels to ~0: fp32 noise, held to the
            # largest gradient instead)
            assert err < 2e-3 or diff < 2e-5 * scale, (train, name, err)


def _structured_grad_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from haet_amd import _autograd
    from haet_amd.PhysicsAttention import Physics_Attention_Structured_Mesh_2D, Physics_Attention_Structured_Mesh_3D
    from haet_amd.sharding import shard_bounds, shard_points

    dev = torch.device("cuda:0")
    out = {}
    for kind, grid in (("structured2d", (12, 10)), ("structured3d", (6, 5, 4))):
        torch.manual_seed(17)
        if kind == "structured2d":
            mod = Physics_Attention_Structured_Mesh_2D(128, heads=4, dim_head=32, slice_num=32, H=grid[0], W=grid[1], rotate=45)
        else:
            mod = Physics_Attention_Structured_Mesh_3D(128, heads=4, dim_head=32, slice_num=32, H=grid[0], W=grid[1],
                                                       D=grid[2], rotate=45)
        with torch.no_grad():
            for name, p in mod.erwin.named_parameters():
                if p.dim() >= 2:
                    p.mul_(3.0)
        mod = mod.to(dev).eval()
        N = 1
        for g in grid:
            N *= g
        plane = N // grid[0]
        x = torch.randn(2, N, 128).to(dev)
        gout = torch.randn(2, N, 128).to(dev)
        xr = x.clone().requires_grad_(True)
        (_autograd.attention_forward(mod, xr) * gout).sum().backward()   # unsharded reference on this rank
        ref = {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}
        ref_gx = xr.grad.clone()
        mod.zero_grad(set_to_none=True)
        shard_points(mod)
        a, b = shard_bounds(grid[0], rank, world)
        xs = x[:, a * plane:b * plane].contiguous().requires_grad_(True)
        (mod(xs) * gout[:, a * plane:b * plane]).sum().backward()
        out[kind] = dict(gx=xs.grad.cpu(), ref_gx=ref_gx[:, a * plane:b * plane].cpu(),
                         grads={n: p.grad.cpu() for n, p in mod.named_parameters() if p.grad is not None},
                         ref={n: g.cpu() for n, g in ref.items()})
    torch.save(out, os.path.join(tmp, f"sg{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_slab_sharded_gradients(tmp_path, hip_device):
    """Gradients on the slab-sharded structured meshes: the convolution's backward on the halo-extended slab and
    the reverse halo exchange; input gradients are slices of the unsharded ones, parameter gradients additive shares."""
    from oracle import haet_oracle as O

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_structured_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(os.path.join(tmp_path, f"sg{r}.pt")) for r in range(2)]
    for kind in ("structured2d", "structured3d"):
        for r in range(2):
            assert O.rel_l2(res[r][kind]["gx"], res[r][kind]["ref_gx"]) < 1e-3, (kind, r)
        ref = res[0][kind]["ref"]
        scale = max(float(g.norm()) for g in ref.values())
        for name, g_ref in ref.items():
            total = res[0][kind]["grads"][name] + res[1][kind]["grads"][name]
            err = float((total - g_ref).norm()) / max(float(g_ref.norm()), 1e-4 * scale)
            assert err < 2e-3, (kind, name, err)
