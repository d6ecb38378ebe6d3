"""Differentiable path of the PhysicsAttention modules.

`loss.backward()` through `models.PhysicsAttention.*` (exp_airfoil.py:249-255) runs as ONE autograd
node, `PhysicsAttentionFunction`, whose N-scaled work is all on the HIP kernels:

    forward   projection GEMM / conv -> slice pass -> merge -> Erwin -> de-slice -> to_out GEMM
    backward  g_out = gy . W_out (GEMM) -> haet_deslice_bwd_tokens (dS2) -> Erwin backward (dS)
              -> haet_slice_bwd_points (dx_lin + the Rep-Slice / Ada-Temp parameter gradients)
              -> gx = dx_lin . W_in (GEMM); the two weight gradients are the split-bf16 G^T.X kernel (csrc/gemm_tn.hip)

The [B,H,N,M] slice-weight tensor is never materialised in either direction.  The Erwin stage -- T = B*H*M tokens
(512 at the 1M-point config), independent of N -- runs forward on the HIP kernel with a tape and backward on the
hand-written `haet_erwin_bwd` (one launch; `ErwinTransformer.forward_taped` / `backward_hip`) wherever one cloud fits that
kernel's LDS plan (`haet_erwin_bwd_supported`: clouds of <= 64 tokens at 32 channels, 128 tokens at 16 channels).  Outside
it (C5's 128 tokens x 32 channels) the stage differentiates through this file's torch restatement (`erwin_forward`,
against models/components/erwinflash/*): in eval mode the forward uses the Erwin HIP kernel and the backward re-runs the
restatement on the saved tokens; in train mode (BatchNorm batch statistics over all clouds, running-stat updates) the
restatement runs in the forward and its graph is kept.  The ball tree comes from the HIP builder either way
(`ops.balltree_build`, bit-identical to the reference's host builder): no CPU round trip, no path
without libhaet_hip.so.  `attention_forward` is the whole module as stock torch ops on the device --
the A/B reference of the tests (and `HipForwardTorchBackward`, the previous interim backward).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops


# --------------------------------------------------------------------------- Erwin (erwin_flash.py:138-195)
def _rms_norm(x, weight):
    # nn.RMSNorm(dim) with eps=None -> finfo(dtype).eps (components/layers.py:25-26); the fused op is one
    # kernel forward and one backward instead of ~six and ~twelve (the stage is launch-bound)
    return F.rms_norm(x, (x.shape[-1],), weight, None)


def _ball_msa(att, x, pos):
    """components/attention.py:42-91 in fp32, no distance bias (the flash variant never applies sigma_att)."""
    ball, heads = att.ball_size, att.num_heads
    p = pos.view(-1, ball, pos.shape[1])
    rel = (p - p.mean(dim=1, keepdim=True)).reshape(-1, pos.shape[1])
    x = x + att.pe_proj(rel)
    c = x.shape[1]
    hd = c // heads
    # one permuted copy for q, k and v: [3, balls, heads, ball, hd] (the stage is launch-bound: every op counts)
    qkv = att.qkv(x).view(-1, ball, 3, heads, hd).permute(2, 0, 3, 1, 4).contiguous()
    o = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])  # softmax(q k^T / sqrt(hd)) v, one autograd node
    return att.proj(o.permute(0, 2, 1, 3).reshape(-1, c))


def _block(blk, x, pos):
    """components/layers.py:30-32."""
    x = x + _ball_msa(blk.BMSA, _rms_norm(x, blk.norm1.weight), pos)
    h = _rms_norm(x, blk.norm2.weight)
    return x + blk.swiglu.w3(blk.swiglu.w2(h) * F.silu(blk.swiglu.w1(h)))


def _run_blocks(layer, x, pos, rot):
    """components/layers.py:72-92: odd blocks on the rotated permutation, un-permuted by argsort."""
    use_rot = rot is not None and len(layer.blocks) > 1
    inv = torch.argsort(rot) if use_rot else None
    for j, blk in enumerate(layer.blocks):
        if use_rot and j % 2 == 1:
            x = _block(blk, x[rot], pos[rot])[inv]
        else:
            x = _block(blk, x, pos)
    return x


def erwin_forward(er, feat, pos, n_clouds):
    """ErwinTransformer.forward with embed=False, mp_steps=0; BatchNorm follows er.training."""
    T = feat.shape[0]
    M = T // n_clouds
    with torch.no_grad():
        tree_idx, tree_mask, rot_flat, sizes = ops.balltree_build(
            pos.detach(), n_clouds, M, er.strides, er.ball_sizes, float(er.rotate))
    tree_idx = tree_idx.long()
    tree_mask = tree_mask.bool() if ops.tree_slots(M) != M else None
    rots, off = [], 0
    for sz in sizes:
        rots.append(None if rot_flat is None else rot_flat[off:off + sz].long())
        off += sz
    d = pos.shape[1]
    x, p = feat[tree_idx], pos[tree_idx]
    stack = []
    for i, layer in enumerate(er.encoder):
        x = _run_blocks(layer, x, p, rots[i])
        stack.append((x, p, rots[i]))
        st = layer.pool.stride  # components/pooling.py:25-40
        if st == 1:  # "no pooling" (pooling.py:26-27): the node passes through, no projection / BatchNorm
            continue
        centers = p.view(-1, st, d).mean(dim=1)
        rel = (p.view(-1, st, d) - centers[:, None]).reshape(-1, st * d)
        x = layer.pool.norm(layer.pool.proj(torch.cat([x.reshape(-1, st * x.shape[1]), rel], dim=1)))
        p = centers
    x = _run_blocks(er.bottleneck, x, p, rots[len(er.encoder)])
    for k, layer in enumerate(er.decoder):
        cx, cp, crot = stack[len(er.encoder) - 1 - k]
        st = layer.unpool.stride  # components/pooling.py:58-72
        rel = (cp.view(-1, st, d) - p[:, None]).reshape(-1, st * d)
        y = layer.unpool.proj(torch.cat([x, rel], dim=-1)).reshape(cx.shape[0], -1)
        x, p = layer.unpool.norm(cx + y), cp
        x = _run_blocks(layer, x, p, crot)
    if ops.tree_slots(M) == M:  # every slot holds a token: no mask, static shapes (HIP-graph capturable)
        return x[torch.argsort(tree_idx)]
    return x[tree_mask][torch.argsort(tree_idx[tree_mask])]


# --------------------------------------------------------------------------- PhysicsAttention (IrregularMesh.py:141-235)
def attention_forward(mod, x):
    B, N, _ = x.shape
    H, C, G = mod.heads, mod.dim_head, mod.slice_num
    x_lin = mod.project_torch(x).float()
    xp = x_lin.reshape(B, N, H, C).permute(0, 2, 1, 3)
    tau = mod.base_temp + torch.clamp(mod.ada_temp_linear(xp), -0.4, 0.4)
    c0 = torch.log(-torch.log(torch.tensor(mod.epsilon, dtype=torch.float32, device=x.device)))
    w = torch.softmax((mod.in_project_slice(xp) - c0) / tau, dim=2)      # over the POINT axis
    S = torch.einsum("bhnc,bhng->bhgc", xp, w) / (w.sum(2) + 1e-5)[..., None]
    flat = S.reshape(B * H * G, C)
    d = mod.dimensionality
    lo, hi = flat[:, :d].min(dim=0)[0], flat[:, :d].max(dim=0)[0]
    pos = ((flat[:, :d] - lo) / (hi - lo + 1e-8)).detach()  # no gradient through positions (see _tokens_through_erwin)
    S2 = erwin_forward(mod.erwin, flat, pos, B * H).reshape(B, H, G, C)
    out_x = torch.einsum("bhgc,bhng->bhnc", S2, w).permute(0, 2, 1, 3).reshape(B, N, H * C)
    return mod.to_out(out_x).to(x.dtype)


class HipForwardTorchBackward(torch.autograd.Function):
    """Forward on the HIP kernels, backward through the torch restatement (eval mode only: the two
    forwards agree to ~1e-6, so the gradient is that of the function the kernels compute)."""

    @staticmethod
    def forward(ctx, mod, x, *params):
        ctx.mod, ctx.all_params = mod, params
        ctx.save_for_backward(x)
        with torch.no_grad():
            return mod._forward_hip(x)

    @staticmethod
    def backward(ctx, grad_out):
        mod = ctx.mod
        (x,) = ctx.saved_tensors
        params = [p for p, need in zip(ctx.all_params, ctx.needs_input_grad[2:]) if need]
        with torch.enable_grad():
            xr = x.detach().requires_grad_(ctx.needs_input_grad[1])
            out = attention_forward(mod, xr)
            wanted = ([xr] if xr.requires_grad else []) + params
            grads = torch.autograd.grad(out, wanted, grad_out, allow_unused=True) if wanted else []
        gi = iter(grads)
        gx = next(gi) if xr.requires_grad else None
        by_param = {id(p): g for p, g in zip(params, gi)}
        return (None, gx) + tuple(by_param.get(id(p)) for p in ctx.all_params)


# --------------------------------------------------------------------------- the HIP training path
def _tokens_through_erwin(mod, S, n_clouds):
    """Token positions (IrregularMesh.py:196-209) + Erwin on the slice tokens S [T, C], differentiable."""
    d = mod.dimensionality
    lo, hi = S[:, :d].min(dim=0)[0], S[:, :d].max(dim=0)[0]
    # The reference uses positions only under no_grad (attention.py:35 `compute_rel_pos`, pooling.py:29,59,
    # the tree build erwin_flash.py:149): nothing flows S -> pos -> pe_proj / pool / unpool.
    pos = ((S[:, :d] - lo) / (hi - lo + 1e-8)).detach()
    return erwin_forward(mod.erwin, S, pos, n_clouds)


class PhysicsAttentionFunction(torch.autograd.Function):
    """forward(x) up to and including the to_out Linear (dropout stays outside), see the module docstring."""

    @staticmethod
    def forward(ctx, mod, x, *params):
        B, N, _ = x.shape
        H, C, G = mod.heads, mod.dim_head, mod.slice_num
        sp = mod._slice_params()
        mod.__dict__["_event_chain"] = [None] if mod.stage_events is not None else None
        ev = mod._stage
        with ev("project"):
            x_lin = mod.project(x).float().contiguous()                  # [B, N, H*C] fp32, kept for backward
            # dim_head 16: the forward's passes run at 32 channels per head, the upper 16 zero
            # (PhysicsAttentionBase.head_pad_applies) -- and, from HEAD_PAD_BWD_ROWS rows on (tools/head_pad_probe.py),
            # the backward's too: then the PADDED x_lin is what is kept for it
            fwd_padded = mod.head_pad_applies(B, N)
            ctx.padded = fwd_padded and B * N >= mod.HEAD_PAD_BWD_ROWS
            Cr, x_real = C, x_lin
            if fwd_padded:
                x_lin, sp, C = ops.pad_heads(x_lin, H, C, mod.HEAD_PAD), mod._slice_params_padded(), mod.HEAD_PAD
        with ev("slice"):
            parts = ops.slice_partials(x_lin, sp, H, C, G)
        with ev("merge"):
            if mod.shard_group is not None:  # point-sharded: one all-gather of the per-rank records
                from .sharding import allgather_records

                parts = allgather_records(ops.merge_records(parts, B, H, C, G, finalize=False), mod.shard_group)
            col_max, col_sum, S = ops.merge_records(parts, B, H, C, G, finalize=True)
            if fwd_padded:
                S = S[:, :Cr].contiguous()
        ctx.inner = None
        ctx.erwin_ctx = None
        with ev("erwin"):
            if mod.erwin.hip_backward_supported(B * H, G):
                # the stage on the HIP kernel in either mode (train: BatchNorm on batch statistics over all clouds,
                # running statistics updated), with the tape its hand-written backward (haet_erwin_bwd) reads
                S2, ctx.erwin_ctx = mod.erwin.forward_taped(S, B * H)
            elif mod.training:  # batch-statistics BatchNorm: the restatement runs now and keeps its graph
                with torch.enable_grad(), torch.autocast("cuda", enabled=False):
                    S_leaf = S.detach().requires_grad_(True)
                    S2_graph = _tokens_through_erwin(mod, S_leaf, B * H)
                S2 = S2_graph.detach()
                ctx.inner = (S_leaf, S2_graph)
            else:
                S2 = mod.erwin(S, None, n_clouds=B * H)
        with ev("deslice"):
            S2p = F.pad(S2, (0, C - Cr)) if fwd_padded else S2
            out_x = ops.deslice(x_lin, sp, col_max, col_sum, S2p, H, C, G)
            if fwd_padded:
                out_x = ops.unpad_heads(out_x, H, Cr, C)
            if not ctx.padded:
                x_lin = x_real  # (the backward runs at the real width)
        with ev("to_out"):
            y = mod._linear("to_out", mod.to_out[0], out_x)
        ctx.save_for_backward(x, x_lin, col_max, col_sum, S, S2, out_x)
        ctx.mod, ctx.all_params = mod, params
        return y

    @staticmethod
    def backward(ctx, gy):
        mod = ctx.mod
        x, x_lin, col_max, col_sum, S, S2, out_x = ctx.saved_tensors
        B, N, _ = x.shape
        H, C, G = mod.heads, mod.dim_head, mod.slice_num
        sp = mod._slice_params()
        Cr = C
        if ctx.padded:  # (x_lin was saved padded; S, S2 and out_x at the real width)
            sp, C = mod._slice_params_padded(), mod.HEAD_PAD
        mod.__dict__["_event_chain"] = [None] if mod.stage_events is not None else None
        ev = mod._stage  # optional per-stage HIP events (tools/train_probe.py)
        wanted = {id(p) for p, need in zip(ctx.all_params, ctx.needs_input_grad[2:]) if need}
        grads = {}
        gy = gy.float().contiguous()
        gy2 = gy.view(B * N, -1)
        lin_out = mod.to_out[0]
        with ev("bwd_to_out"):
            if id(lin_out.weight) in wanted or (lin_out.bias is not None and id(lin_out.bias) in wanted):
                grads[id(lin_out.weight)], gb = ops.linear_wgrad(gy2, out_x.view(B * N, -1))
                if lin_out.bias is not None:
                    grads[id(lin_out.bias)] = gb
            g_out = mod._linear_transposed("to_out", lin_out, gy)       # dL/d out_x [B, N, H*C]
            if ctx.padded:
                g_out = ops.pad_heads(g_out, H, Cr, C)
        with ev("bwd_tokens"):
            dS2 = ops.deslice_bwd_tokens(x_lin, sp, col_max, col_sum, g_out, H, C, G)
            if ctx.padded:
                dS2 = dS2[:, :Cr].contiguous()
            share = 1.0
            if mod.shard_group is not None:  # the backward's one exchange; everything token-level is replicated
                import torch.distributed as dist

                from .sharding import allreduce_sum

                allreduce_sum(dS2, mod.shard_group)
                share = 1.0 / dist.get_world_size(mod.shard_group)

        with ev("bwd_erwin"):
            erwin_params = [p for p in mod.erwin.param_list() if id(p) in wanted]
            if ctx.erwin_ctx is not None:  # hand-written backward of the whole stage, one launch
                dS, by_param = mod.erwin.backward_hip(ctx.erwin_ctx, dS2)
                ctx.erwin_ctx = None
                got = [dS] + [by_param.get(id(p)) for p in erwin_params]
            elif ctx.inner is None:
                with torch.enable_grad(), torch.autocast("cuda", enabled=False):
                    S_leaf = S.detach().requires_grad_(True)
                    S2_graph = _tokens_through_erwin(mod, S_leaf, B * H)
                got = torch.autograd.grad(S2_graph, [S_leaf] + erwin_params, dS2, allow_unused=True)
            else:
                S_leaf, S2_graph = ctx.inner
                got = torch.autograd.grad(S2_graph, [S_leaf] + erwin_params, dS2, allow_unused=True)
            ctx.inner = None
            dS = got[0] if got[0] is not None else torch.zeros_like(S)
            for p, g in zip(erwin_params, got[1:]):  # replicated: each rank reports its additive share (sharding.py)
                grads[id(p)] = g if share == 1.0 or g is None else g * share
            A = (dS2 * S2).sum(dim=-1) + (S * dS).sum(dim=-1)          # sum_n w[n,g] a[n,g], [B*H*M]
        with ev("bwd_points"):
            if ctx.padded:
                S2, dS = F.pad(S2, (0, C - Cr)), F.pad(dS, (0, C - Cr))
            dx_lin, dWs, dbs, dwt, dbt = ops.slice_bwd_points(x_lin, sp, col_max, col_sum, g_out, S2, dS, A, H, C, G)
            if ctx.padded:
                dx_lin, dWs, dwt, C = ops.unpad_heads(dx_lin, H, Cr, C), dWs[:, :Cr], dwt.reshape(-1)[:Cr], Cr
        grads[id(mod.in_project_slice.weight)] = dWs
        grads[id(mod.in_project_slice.bias)] = dbs
        grads[id(mod.ada_temp_linear.weight)] = dwt.view(1, C)
        grads[id(mod.ada_temp_linear.bias)] = dbt.view(1)
        with ev("bwd_project"):
            gx, proj_grads = mod.project_backward(x, dx_lin, ctx.needs_input_grad[1], wanted)
        grads.update(proj_grads)
        if gx is not None:
            gx = gx.to(x.dtype)
        out = []
        for p in ctx.all_params:
            g = grads.get(id(p)) if id(p) in wanted else None
            out.append(None if g is None else g.to(p.dtype).reshape(p.shape))
        return (None, gx) + tuple(out)
