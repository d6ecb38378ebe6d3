"""Pack an ErwinTransformer's parameters into the single fp32 blob + descriptor that
haet_erwin_fwd consumes (layout contract: include/haet_hip.h, haet_erwin_desc).

Linear weights keep PyTorch's [out][in] layout with rows zero-padded to a multiple of 32
floats (one lane of the MFMA B operand reads 8 consecutive k of one output row); eval-mode BatchNorm1d is folded to y*scale + shift
(scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale;
models/components/erwinflash/components/pooling.py:23,38,56,68)."""
from __future__ import annotations

import torch

from ..._lib import MAX_DEPTH, MAX_LEVELS, ErwinDesc, HaetError


class _Blob:
    def __init__(self):
        self.chunks, self.n = [], 0

    def add(self, t: torch.Tensor) -> int:
        t = t.detach().float().reshape(-1)
        pad = (-self.n) % 4  # keep every tensor 16-byte aligned
        if pad:
            self.chunks.append(torch.zeros(pad, device=t.device))
            self.n += pad
        off = self.n
        self.chunks.append(t)
        self.n += t.numel()
        return off

    def lin(self, linear):
        """[out][in] rows zero-padded to a multiple of 32 floats (MFMA k-step)."""
        w = linear.weight.detach().float()
        pad = (-w.shape[1]) % 32
        if pad:
            w = torch.nn.functional.pad(w, (0, pad))
        return self.add(w.contiguous()), self.add(linear.bias)

    def lin_t(self, linear):
        """transposed [in][out] (the 2/3-wide positional projection, evaluated on the VALU)."""
        return self.add(linear.weight.t().contiguous()), self.add(linear.bias)

    def tensor(self):
        return torch.cat(self.chunks).contiguous()


def _fill_layer(dst, layer, blob):
    dst.c, dst.heads, dst.ball = layer.hidden_dim, layer.num_heads, layer.ball_size
    dst.depth, dst.hidden = len(layer.blocks), layer.hidden_dim * layer.mlp_ratio
    if dst.depth > MAX_DEPTH:
        raise HaetError(f"Erwin layer depth {dst.depth} > {MAX_DEPTH}")
    for j, blk in enumerate(layer.blocks):
        w = dst.blocks[j]
        w.norm1, w.norm2 = blob.add(blk.norm1.weight), blob.add(blk.norm2.weight)
        w.pe_w, w.pe_b = blob.lin_t(blk.BMSA.pe_proj)
        w.qkv_w, w.qkv_b = blob.lin(blk.BMSA.qkv)
        w.proj_w, w.proj_b = blob.lin(blk.BMSA.proj)
        w.w1_w, w.w1_b = blob.lin(blk.swiglu.w1)
        w.w2_w, w.w2_b = blob.lin(blk.swiglu.w2)
        w.w3_w, w.w3_b = blob.lin(blk.swiglu.w3)


def _fill_resample(dst, mod, blob):
    dst.stride = mod.stride
    dst.w, dst.b = blob.lin(mod.proj)
    bn = mod.norm
    scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
    dst.bn_scale = blob.add(scale)
    dst.bn_shift = blob.add(bn.bias.detach() - bn.running_mean * scale)


def pack_erwin(erwin):
    n_levels = len(erwin.encoder) + 1
    if n_levels > MAX_LEVELS:
        raise HaetError(f"Erwin with {n_levels} levels > {MAX_LEVELS}")
    desc, blob = ErwinDesc(), _Blob()
    desc.n_levels, desc.d = n_levels, erwin.dimensionality
    desc.rotate, desc.decode = int(erwin.rotate > 0), int(bool(erwin.decode))
    for i, layer in enumerate(erwin.encoder):
        _fill_layer(desc.enc[i], layer, blob)
        _fill_resample(desc.pool[i], layer.pool, blob)
    _fill_layer(desc.enc[n_levels - 1], erwin.bottleneck, blob)
    # decoder[k] runs at level n_levels-2-k (erwin_flash.py:108-123)
    for k, layer in enumerate(erwin.decoder):
        lvl = n_levels - 2 - k
        _fill_layer(desc.dec[lvl], layer, blob)
        _fill_resample(desc.unpool[lvl], layer.unpool, blob)
    return blob.tensor(), desc
