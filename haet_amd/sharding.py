"""Point-sharded multi-GPU execution of the PhysicsAttention path.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU
tests).  Every rank holds a shard of the N points; the ONLY exchange per forward is the
all-gather of the per-rank slice record  [B*H*M*(C+2)] floats  (~70 KB at C3), merged by
the log-sum-exp rule on every rank.  The message is latency-bound, so it is a single
all-gather (one hop on the fully connected xGMI node), not a ring all-reduce of max + sum.
Everything after the merge (token positions, ball tree, Erwin) is replicated.

Gradients (Physics_Attention_Irregular_Mesh): the backward mirrors it with ONE all-reduce of the partial
dS2 [B*H*M, C] (64 KB at C3); the Erwin backward is replicated.  Every rank then holds its ADDITIVE SHARE of
each parameter gradient -- the sums over its own points for in_project_x / in_project_slice / ada_temp_linear /
to_out, and 1/R of the (replicated) Erwin gradients -- so that the trainer's ordinary SUM all-reduce over all
parameters (the same one the point-wise layers around the module need) yields the full gradient.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def allgather_records(local_record: torch.Tensor, group=None) -> torch.Tensor:
    """[rec] on each rank -> [R, rec] in rank order on every rank."""
    world = dist.get_world_size(group)
    flat = local_record.contiguous().reshape(-1)
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        # test configuration only (several ranks sharing one GPU): gloo moves host memory
        host = torch.empty(world * flat.numel(), dtype=flat.dtype)
        dist.all_gather_into_tensor(host, flat.cpu(), group=group)
        return host.to(flat.device).view(world, flat.numel())
    out = torch.empty(world * flat.numel(), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)  # flat output: accepted by RCCL and gloo
    return out.view(world, flat.numel())


def allreduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks (the backward's one exchange: the partial dS2 of every shard)."""
    if t.is_cuda and dist.get_backend(group) == "gloo":  # test configuration, see allgather_records
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_gradients(model: torch.nn.Module, group=None) -> None:
    """The trainer-side half of sharded training: SUM all-reduce of every parameter gradient over the shard group,
    in ONE bucket (a few MB per model: one ring pass over xGMI instead of one latency per tensor).  Afterwards every
    rank holds the full gradient of the summed loss -- the point-wise layers' local sums and this package's additive
    shares alike (see the module docstring).  Call between loss.backward() and optimizer.step()."""
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    allreduce_sum(flat, group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def shard_points(model: torch.nn.Module, group=None) -> torch.nn.Module:
    """Mark every Physics_Attention_* module of `model` as point-sharded over `group`."""
    from .PhysicsAttention._base import PhysicsAttentionBase

    for m in model.modules():
        if isinstance(m, PhysicsAttentionBase):
            m.shard_group = group if group is not None else dist.group.WORLD
    return model


def shard_bounds(n_points: int, rank: int, world: int):
    """Contiguous, balanced split of n_points (first n % world ranks get one extra)."""
    base, extra = divmod(n_points, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def exchange_planes(send_prev, send_next, group=None):
    """Point-to-point exchange with the two neighbour ranks of a slab decomposition: `send_prev` goes to rank - 1,
    `send_next` to rank + 1 (either may be None at the global faces); returns (what rank - 1 sent as its
    `send_next`, what rank + 1 sent as its `send_prev`), None at the faces.  ncclSend/Recv over xGMI on the GPU node."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ref = send_prev if send_prev is not None else send_next
    if ref is None:
        return None, None
    via_host = ref.is_cuda and dist.get_backend(group) == "gloo"  # test configuration (see allgather_records)
    buf = (lambda: torch.empty(ref.shape, dtype=ref.dtype, device="cpu" if via_host else ref.device))
    prev = buf() if rank > 0 else None
    nxt = buf() if rank < world - 1 else None
    ops = []
    peer = (lambda r: dist.get_global_rank(group, r) if group is not None else r)
    out = (lambda t: t.contiguous().cpu() if via_host else t.contiguous())
    if rank > 0:
        ops += [dist.P2POp(dist.isend, out(send_prev), peer(rank - 1), group),
                dist.P2POp(dist.irecv, prev, peer(rank - 1), group)]
    if rank < world - 1:
        ops += [dist.P2POp(dist.isend, out(send_next), peer(rank + 1), group),
                dist.P2POp(dist.irecv, nxt, peer(rank + 1), group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if via_host:
        prev = None if prev is None else prev.to(ref.device)
        nxt = None if nxt is None else nxt.to(ref.device)
    return prev, nxt


def exchange_halo_planes(x: torch.Tensor, plane: int, group=None):
    """Slab sharding of a structured grid along its first axis: x [B, n_planes*plane, C]
    is this rank's slab; returns (prev, next) = the neighbour ranks' boundary planes [B, plane, C]
    (None at the global faces, where the convolution's zero padding applies).  One point-to-point
    exchange per side."""
    n = x.shape[1]
    return exchange_planes(x[:, :plane], x[:, n - plane:], group)
