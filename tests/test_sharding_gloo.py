"""CPU, world_size = 2, gloo: the N > 1 path.  The exchange code is the product's
(haet_amd.sharding.allgather_records / shard_bounds); the arithmetic on each rank is the CPU
oracle's (the HIP kernels need a GPU), so this checks that gathering per-rank log-sum-exp
records in rank order and merging them reproduces the unsharded eidetic states."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _record(x_lin, sd, cfg):
    """Per-rank record [B*H][ m[M] | l[M] | St[M][C] ] (layout of include/haet_hip.h) via the oracle."""
    from oracle import haet_oracle as O

    B, N, _ = x_lin.shape
    H, C, M = cfg.heads, cfg.dim_head, cfg.slice_num
    xp = x_lin.reshape(B, N, H, C).permute(0, 2, 1, 3)
    tau, z = O.slice_logits(xp, sd, cfg)
    m = z.max(dim=2)[0]                                  # [B,H,M]
    p = torch.exp(z - m[:, :, None, :])
    l = p.sum(2)
    St = torch.einsum("bhnc,bhng->bhgc", xp, p)
    return torch.cat([m, l, St.reshape(B, H, M * C)], dim=2).reshape(-1)


def _merge(records, B, H, C, M):
    r = records.reshape(records.shape[0], B * H, M * (C + 2))
    m, l, St = r[..., :M], r[..., M:2 * M], r[..., 2 * M:].reshape(-1, B * H, M, C)
    mx = m.max(dim=0)[0]
    a = torch.exp(m - mx)
    lsum = (l * a).sum(0)
    S = (St * a[..., None]).sum(0) / lsum[..., None] / (1.0 + 1e-5)
    return S.reshape(-1, C)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from haet_amd.sharding import allgather_records, shard_bounds
    from oracle import haet_oracle as O

    torch.manual_seed(0)
    B, N, H, C, M = 2, 1001, 4, 16, 8
    x_lin = torch.randn(B, N, H * C)
    sd = {"in_project_slice.weight": torch.randn(M, C), "in_project_slice.bias": torch.randn(M),
          "ada_temp_linear.weight": torch.randn(1, C) * 0.3, "ada_temp_linear.bias": torch.tensor([0.1])}
    cfg = O.AttnCfg(dim=H * C, heads=H, dim_head=C, slice_num=M)
    a, b = shard_bounds(N, rank, world)
    local = _record(x_lin[:, a:b].contiguous(), sd, cfg)
    gathered = allgather_records(local, None)
    assert gathered.shape == (world, B * H * M * (C + 2))
    assert torch.equal(gathered[rank], local)
    S = _merge(gathered, B, H, C, M)
    S_ref = _merge(_record(x_lin, sd, cfg)[None], B, H, C, M)
    err = O.rel_l2(S, S_ref)
    torch.save({"err": err}, os.path.join(tmp, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_allgather_merge_world2(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert torch.load(os.path.join(tmp_path, f"r{r}.pt"))["err"] < 1e-5


def test_shard_bounds_cover_every_point_once():
    from haet_amd.sharding import shard_bounds

    for n in (1, 7, 8, 1000, 1_000_003):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _halo_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from haet_amd.sharding import exchange_halo_planes, shard_bounds

    torch.manual_seed(0)
    B, gH, plane, C = 2, 7, 6, 5                    # 7 planes of 6 points over 3 ranks: 3 + 2 + 2
    full = torch.randn(B, gH * plane, C)
    a, b = shard_bounds(gH, rank, world)
    prev, nxt = exchange_halo_planes(full[:, a * plane:b * plane].contiguous(), plane, None)
    ok = True
    ok &= (prev is None) == (rank == 0) and (nxt is None) == (rank == world - 1)
    if prev is not None:
        ok &= torch.equal(prev, full[:, (a - 1) * plane:a * plane])
    if nxt is not None:
        ok &= torch.equal(nxt, full[:, b * plane:(b + 1) * plane])
    torch.save({"ok": bool(ok)}, os.path.join(tmp, f"h{rank}.pt"))
    dist.destroy_process_group()


def test_halo_plane_exchange_three_ranks(tmp_path):
    """Slab sharding of the structured grids: every rank receives its neighbours' boundary planes and
    nothing at the global faces (the exchange the Conv2d / Conv3d projection needs)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_halo_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        assert torch.load(os.path.join(tmp_path, f"h{r}.pt"))["ok"]


def _grad_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from haet_amd.sharding import allreduce_gradients, exchange_planes, shard_bounds

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.GELU(), torch.nn.Linear(5, 3))
    x = torch.randn(40, 6)
    ref = [torch.autograd.grad(model(x).square().sum(), list(model.parameters()))][0]   # the whole batch on one rank
    a, b = shard_bounds(40, rank, world)
    model(x[a:b]).square().sum().backward()                                             # this rank's rows only
    allreduce_gradients(model)
    ok = all(torch.allclose(p.grad, g, atol=1e-5) for p, g in zip(model.parameters(), ref))
    # the reverse halo exchange of the sharded convolution's backward: what I send "up" arrives as my neighbour's "from below"
    up, down = torch.full((1, 2, 3), float(rank)), torch.full((1, 2, 3), 10.0 + rank)
    got_prev, got_next = exchange_planes(up if rank > 0 else None, down if rank < world - 1 else None, None)
    ok &= (got_prev is None) == (rank == 0) and (got_next is None) == (rank == world - 1)
    if got_prev is not None:
        ok &= bool((got_prev == 10.0 + rank - 1).all())
    if got_next is not None:
        ok &= bool((got_next == float(rank + 1)).all())
    torch.save({"ok": bool(ok)}, os.path.join(tmp, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_allreduce_and_reverse_exchange_three_ranks(tmp_path):
    """The trainer-side SUM all-reduce of sharded training (every rank ends with the full-batch gradient) and the
    neighbour exchange in the direction the convolution's backward uses it."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_grad_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        assert torch.load(os.path.join(tmp_path, f"g{r}.pt"))["ok"]
