"""Inference fast path of the point-wise layers AROUND the attention: the LayerNorms and the
Linear (+ GELU) layers of a Transolver block and of the preprocess MLP (models/HAETransolver_Irregular_Mesh.py:
114-134, 186, 338-376; models/components/mlp.py) on the HIP kernels -- the split-bf16 MFMA GEMM with the GELU in
its epilogue, a one-pass LayerNorm, a one-pass narrow Linear.  At 1M points these stock-torch layers cost 13 of
the 15 ms of a one-layer model forward, the attention itself 1.8.

Only with gradients disabled, fp32 tensors on the device; anything else (training, autocast dtypes, shapes
outside the kernels' envelopes, HAET_NO_BLOCK_FAST=1) runs the stock modules unchanged.
"""
from __future__ import annotations

import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

_split = weakref.WeakKeyDictionary()  # nn.Linear -> (weight / bias version key, ops.SplitLinear)


def enabled(x: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and not os.environ.get("HAET_NO_BLOCK_FAST"))


def _split_of(lin: nn.Linear, shift: torch.Tensor = None) -> ops.SplitLinear:
    """Split planes of a Linear, cached per parameter version; `shift` [O] is folded into the bias (a constant
    added to every output row, e.g. the models' `placeholder`)."""
    b = lin.bias
    key = (lin.weight.data_ptr(), lin.weight._version, None if b is None else (b.data_ptr(), b._version),
           None if shift is None else (shift.data_ptr(), shift._version))
    hit = _split.get(lin)
    if hit is None or hit[0] != key:
        bias = b
        if shift is not None:
            bias = shift.detach().float() if b is None else b.detach().float() + shift.detach().float()
        hit = (key, ops.SplitLinear(lin.weight, bias))
        _split[lin] = hit
    return hit[1]


def linear(lin: nn.Linear, x: torch.Tensor, gelu: bool = False, shift: torch.Tensor = None) -> torch.Tensor:
    K, O = lin.in_features, lin.out_features
    if ops.linear_supported(K, O):
        return ops.linear(x, _split_of(lin, shift), gelu)
    if ops.linear_narrow_supported(K, O):
        y = ops.linear_narrow(x, lin.weight, lin.bias, gelu)
    else:
        y = lin(x)
        y = F.gelu(y) if gelu else y
    return y if shift is None else y.add_(shift)


def mlp(m, x: torch.Tensor, shift: torch.Tensor = None):
    """components.MLP forward (GELU variant) on the HIP kernels, or None when it does not apply;
    `shift` is added to the output (folded into the last bias)."""
    if not (enabled(x) and is_exact_gelu(m.linear_pre[1])):
        return None
    x = linear(m.linear_pre[0], x, gelu=True)
    for i in range(m.n_layers):
        y = linear(m.linears[i][0], x, gelu=True)
        x = y + x if m.res else y
    return linear(m.linear_post, x, shift=shift)


def layer_norm(ln: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
    if (ln.elementwise_affine and ln.bias is not None and len(ln.normalized_shape) == 1
            and ops.layernorm_supported(ln.normalized_shape[0])):
        return ops.layernorm(x, ln.weight, ln.bias, ln.eps)
    return ln(x)


def is_exact_gelu(act: nn.Module) -> bool:
    return isinstance(act, nn.GELU) and getattr(act, "approximate", "none") == "none"
