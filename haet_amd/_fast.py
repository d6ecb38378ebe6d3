"""Inference fast path of the point-wise layers AROUND the attention: the LayerNorms and the
Linear (+ GELU) layers of a Transolver block and of the preprocess MLP (models/HAETransolver_Irregular_Mesh.py:
114-134, 186, 338-376; models/components/mlp.py) on the HIP kernels -- the split-bf16 MFMA GEMM with the GELU in
its epilogue, a one-pass LayerNorm, a one-pass narrow Linear.  At 1M points these stock-torch layers cost 13 of
the 15 ms of a one-layer model forward, the attention itself 1.8.

With gradients disabled the GELU rides in the GEMM's epilogue and the models' `placeholder` in the last bias.
With gradients enabled (training scripts: loss.backward(), exp_airfoil.py:249-255) the same layers run as
autograd nodes on the same kernels -- Linear: forward GEMM, input gradient = the GEMM with the transposed
weight planes, weight / bias gradient = haet_linear_wgrad; LayerNorm: haet_layernorm_fwd / _bwd -- with the
activation left to torch (it keeps the pre-activation for its own backward).  One-layer model, 1M points:
inference 15.1 -> 6.2 ms, training step 58.4 -> 30.6 ms.

fp32 tensors on the device only; anything else (autocast dtypes, shapes outside the kernels' envelopes,
HAET_NO_BLOCK_FAST=1) runs the stock modules unchanged.
"""
from __future__ import annotations

import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops

_split = weakref.WeakKeyDictionary()  # nn.Linear -> (weight / bias version key, ops.SplitLinear)


def usable(x: torch.Tensor) -> bool:
    return x.is_cuda and x.dtype == torch.float32 and not os.environ.get("HAET_NO_BLOCK_FAST")


def enabled(x: torch.Tensor) -> bool:  # the inference flavour (fused epilogues)
    return usable(x) and not torch.is_grad_enabled()


_split_t = weakref.WeakKeyDictionary()  # nn.Linear -> (weight version key, SplitLinear of W^T, no bias)


def _split_t_of(lin: nn.Linear) -> ops.SplitLinear:
    key = (lin.weight.data_ptr(), lin.weight._version, _lib.invalidate_epoch())
    hit = _split_t.get(lin)
    if hit is None or hit[0] != key:
        hit = (key, ops.SplitLinear(lin.weight.detach().t().contiguous(), None))
        _split_t[lin] = hit
    return hit[1]


class _HipLinear(torch.autograd.Function):
    """nn.Linear on the HIP kernels in both directions."""

    @staticmethod
    def forward(ctx, x, weight, bias, lin):
        ctx.lin, ctx.has_bias = lin, bias is not None
        ctx.save_for_backward(x, weight)
        if ops.linear_supported(lin.in_features, lin.out_features):
            return ops.linear(x, _split_of(lin))
        return ops.linear_narrow(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        lin = ctx.lin
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if ops.linear_supported(lin.out_features, lin.in_features):
                gx = ops.linear(gy, _split_t_of(lin))
            else:
                gx = gy @ weight
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = ops.linear_wgrad(gy, x)
        return gx, gw, (gb if ctx.has_bias else None), None


class _HipLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        ctx.eps = eps
        ctx.save_for_backward(x, weight)
        return ops.layernorm(x, weight, bias, eps)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(x, weight, gy.contiguous(), ctx.eps)
        return dx, dg, db, None


def _split_of(lin: nn.Linear, shift: torch.Tensor = None) -> ops.SplitLinear:
    """Split planes of a Linear, cached per parameter version; `shift` [O] is folded into the bias (a constant
    added to every output row, e.g. the models' `placeholder`)."""
    b = lin.bias
    key = (lin.weight.data_ptr(), lin.weight._version, None if b is None else (b.data_ptr(), b._version),
           None if shift is None else (shift.data_ptr(), shift._version), _lib.invalidate_epoch())
    hit = _split.get(lin)
    if hit is None or hit[0] != key:
        bias = b
        if shift is not None:
            bias = shift.detach().float() if b is None else b.detach().float() + shift.detach().float()
        hit = (key, ops.SplitLinear(lin.weight, bias))
        _split[lin] = hit
    return hit[1]


def _ln_fusable(ln, lin: nn.Linear, x: torch.Tensor) -> bool:
    return (ln is not None and not torch.is_grad_enabled() and ln.elementwise_affine and ln.bias is not None
            and tuple(ln.normalized_shape) == (lin.in_features,) and ops.linear_supported(lin.in_features, lin.out_features)
            and ops.linear_ln_supported(lin.in_features, lin.out_features, x.numel() // lin.in_features))


@torch.compiler.disable  # opaque to torch.compile: ctypes calls into libhaet_hip.so
def linear(lin: nn.Linear, x: torch.Tensor, gelu: bool = False, shift: torch.Tensor = None,
           pre_norm: nn.LayerNorm = None) -> torch.Tensor:
    """`pre_norm`: a LayerNorm applied to x first -- inside the GEMM's loader where the kernel allows."""
    K, O = lin.in_features, lin.out_features
    if pre_norm is not None:
        if _ln_fusable(pre_norm, lin, x):
            return ops.linear(x, _split_of(lin, shift), gelu, ln=(pre_norm.weight, pre_norm.bias, pre_norm.eps))
        x = layer_norm(pre_norm, x)
    if torch.is_grad_enabled():  # training: autograd node on the kernels, activation / shift by torch
        if ops.linear_supported(K, O) or ops.linear_narrow_supported(K, O):
            y = _HipLinear.apply(x, lin.weight, lin.bias, lin)
        else:
            y = lin(x)
        y = F.gelu(y) if gelu else y
        return y if shift is None else y + shift
    if ops.linear_supported(K, O):
        return ops.linear(x, _split_of(lin, shift), gelu)
    if ops.linear_narrow_supported(K, O):
        y = ops.linear_narrow(x, lin.weight, lin.bias, gelu)
    else:
        y = lin(x)
        y = F.gelu(y) if gelu else y
    return y if shift is None else y.add_(shift)


@torch.compiler.disable  # opaque to torch.compile: ctypes calls into libhaet_hip.so
def mlp(m, x: torch.Tensor, shift: torch.Tensor = None, pre_norm: nn.LayerNorm = None):
    """components.MLP forward (GELU variant) on the HIP kernels, or None when it does not apply;
    `shift` is added to the output (folded into the last bias), `pre_norm` is applied to x first (in the
    first GEMM's loader where possible)."""
    if not (usable(x) and is_exact_gelu(m.linear_pre[1])):
        return None
    x = linear(m.linear_pre[0], x, gelu=True, pre_norm=pre_norm)
    for i in range(m.n_layers):
        y = linear(m.linears[i][0], x, gelu=True)
        x = y + x if m.res else y
    return linear(m.linear_post, x, shift=shift)


@torch.compiler.disable  # opaque to torch.compile: ctypes calls into libhaet_hip.so
def layer_norm(ln: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
    if (ln.elementwise_affine and ln.bias is not None and len(ln.normalized_shape) == 1
            and ops.layernorm_supported(ln.normalized_shape[0])):
        if torch.is_grad_enabled():
            return _HipLayerNorm.apply(x, ln.weight, ln.bias, ln.eps)
        return ops.layernorm(x, ln.weight, ln.bias, ln.eps)
    return ln(x)


@torch.compiler.disable  # opaque to torch.compile: ctypes calls into libhaet_hip.so
def add_layer_norm(ln: nn.LayerNorm, x: torch.Tensor, res: torch.Tensor):
    """(x + res, ln(x + res)); one kernel at inference."""
    if (enabled(x) and res.shape == x.shape and res.dtype == torch.float32 and ln.elementwise_affine
            and ln.bias is not None and len(ln.normalized_shape) == 1 and ops.layernorm_supported(ln.normalized_shape[0])):
        return ops.add_layernorm(x, res, ln.weight, ln.bias, ln.eps)
    s = x + res
    return s, layer_norm(ln, s)


@torch.compiler.disable  # opaque to torch.compile: ctypes calls into libhaet_hip.so
def add_layer_norm_linear(ln: nn.LayerNorm, lin: nn.Linear, x: torch.Tensor, res: torch.Tensor) -> torch.Tensor:
    """lin(ln(x + res)) -- the model's last layer; one kernel at inference when the output is narrow."""
    if (enabled(x) and res.shape == x.shape and res.dtype == torch.float32 and ln.elementwise_affine and ln.bias is not None
            and tuple(ln.normalized_shape) == (lin.in_features,)
            and ops.ln_linear_small_supported(lin.in_features, lin.out_features)):
        return ops.ln_linear_small(x, res, ln.weight, ln.bias, ln.eps, lin.weight, lin.bias)
    _, h = add_layer_norm(ln, x, res)
    return linear(lin, h)


def is_exact_gelu(act: nn.Module) -> bool:
    return isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none"
