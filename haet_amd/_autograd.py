"""Differentiable path of the PhysicsAttention modules (interim state).

The hand-written HIP kernels are forward-only.  So that the modules still drop into the reference's
training scripts (`loss.backward()` through `models.PhysicsAttention.*`, exp_airfoil.py:249-255):

* eval mode with gradients enabled (fine-tuning with frozen BatchNorm statistics): the forward runs
  on the HIP kernels as always, and `backward` re-runs THIS file's restatement of the same forward
  under autograd on the device and back-propagates through it (`HipForwardTorchBackward`);
* train mode (BatchNorm batch statistics over all clouds, running-stat updates, dropout): forward
  and backward both run here.

Everything below is stock PyTorch on ROCm device tensors, written against the same reference lines
as the kernels (models/PhysicsAttention/IrregularMesh.py:141-235, models/components/erwinflash/*);
the ball tree comes from the HIP builder (`ops.balltree_build`, bit-identical to the reference's
host builder), so there is still no CPU round trip and no path without libhaet_hip.so.  A hand-written
HIP backward replaces this file later; the no-grad / eval forward never comes through here.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import ops


# --------------------------------------------------------------------------- Erwin (erwin_flash.py:138-195)
def _rms_norm(x, weight):
    eps = torch.finfo(x.dtype).eps  # nn.RMSNorm(dim) with eps=None (components/layers.py:25-26)
    return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps) * weight


def _ball_msa(att, x, pos):
    """components/attention.py:42-91 in fp32, no distance bias (the flash variant never applies sigma_att)."""
    ball, heads = att.ball_size, att.num_heads
    p = pos.view(-1, ball, pos.shape[1])
    rel = (p - p.mean(dim=1, keepdim=True)).reshape(-1, pos.shape[1])
    x = x + att.pe_proj(rel)
    c = x.shape[1]
    hd = c // heads
    qkv = att.qkv(x).view(-1, ball, 3, heads, hd)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    w = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(hd)), dim=-1)
    return att.proj(torch.matmul(w, v).permute(0, 2, 1, 3).reshape(-1, c))


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
    tree_idx, tree_mask = tree_idx.long(), tree_mask.bool()
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
    pos = (flat[:, :d] - lo) / (hi - lo + 1e-8)
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
