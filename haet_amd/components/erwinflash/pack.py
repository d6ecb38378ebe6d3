"""Pack an ErwinTransformer's parameters into the single blob + descriptor that
haet_erwin_fwd consumes (layout contract: include/haet_hip.h, haet_erwin_desc).

Blob = [ copy of the descriptor | tensors ], offsets in floats from the blob start.
* Linear weights: PyTorch [out][in] layout, rows zero-padded to Kp = roundup(in, 32), stored
  pre-split as two bf16 planes -- hi = RNE(w) at the offset, lo = RNE(w - hi) O*Kp bf16 later --
  so the kernel feeds them to the MFMA without conversion (3 products hi*hi + hi*lo + lo*hi
  give fp32-class accuracy).
* pe_proj (2/3 inputs) stays fp32, transposed [d][c]; biases / norm weights fp32.
* eval-mode BatchNorm1d is folded to y*scale + shift (scale = gamma / sqrt(running_var + eps),
  shift = beta - running_mean * scale; models/components/erwinflash/components/pooling.py:23,38,56,68);
  in train mode (desc.bn_train) the same two slots carry gamma and beta and the kernel computes the batch
  statistics over all clouds itself.
"""
from __future__ import annotations

import ctypes

import torch

from ..._lib import MAX_DEPTH, MAX_LEVELS, ErwinDesc, HaetError

HEADER_FLOATS = (ctypes.sizeof(ErwinDesc) + 15) // 16 * 4


class _Blob:
    def __init__(self, device):
        self.device = device
        self.chunks, self.n = [], HEADER_FLOATS

    def add(self, t: torch.Tensor) -> int:
        t = t.detach().reshape(-1)
        if t.dtype != torch.float32:
            raise HaetError("blob tensors are appended as fp32 words")
        pad = (-self.n) % 4  # keep every tensor 16-byte aligned
        if pad:
            self.chunks.append(torch.zeros(pad, device=self.device))
            self.n += pad
        off = self.n
        self.chunks.append(t.to(self.device))
        self.n += t.numel()
        return off

    def planes(self, linear):
        """[out][in~32] bf16 hi plane followed by the lo plane (see module docstring)."""
        w = linear.weight.detach().float()
        pad = (-w.shape[1]) % 32
        if pad:
            w = torch.nn.functional.pad(w, (0, pad))
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        planes = torch.cat([hi.reshape(-1), lo.reshape(-1)]).contiguous()
        return self.add(planes.view(torch.float32))

    def end_group(self, start):
        """Close a small-parameter group: pad to 16 bytes, return (offset, length in floats)."""
        pad = (-self.n) % 4
        if pad:
            self.chunks.append(torch.zeros(pad, device=self.device))
            self.n += pad
        return start, self.n - start

    def tensor(self, desc):
        hdr = torch.frombuffer(bytearray(bytes(desc)) + bytearray(HEADER_FLOATS * 4 - ctypes.sizeof(desc)),
                               dtype=torch.float32).to(self.device)
        pad = (-self.n) % 4  # whole 16-byte chunks: the kernel walks the blob with 16-B loads
        tail = [torch.zeros(pad, device=self.device)] if pad else []
        return torch.cat([hdr] + self.chunks + tail).contiguous()


def _fill_layer(dst, layer, blob):
    dst.c, dst.heads, dst.ball = layer.hidden_dim, layer.num_heads, layer.ball_size
    dst.depth, dst.hidden = len(layer.blocks), layer.hidden_dim * layer.mlp_ratio
    if dst.depth > MAX_DEPTH:
        raise HaetError(f"Erwin layer depth {dst.depth} > {MAX_DEPTH}")
    f = lambda t: t.detach().float()
    for j, blk in enumerate(layer.blocks):
        w = dst.blocks[j]
        # small fp32 parameters, contiguous: the kernel copies this group to LDS in one go
        w.norm1 = blob.add(f(blk.norm1.weight))
        w.norm2 = blob.add(f(blk.norm2.weight))
        w.pe_w = blob.add(f(blk.BMSA.pe_proj.weight).t().contiguous())  # [d][c]
        w.pe_b = blob.add(f(blk.BMSA.pe_proj.bias))
        w.qkv_b = blob.add(f(blk.BMSA.qkv.bias))
        w.proj_b = blob.add(f(blk.BMSA.proj.bias))
        w.w1_b = blob.add(f(blk.swiglu.w1.bias))
        w.w2_b = blob.add(f(blk.swiglu.w2.bias))
        w.w3_b = blob.add(f(blk.swiglu.w3.bias))
        w.small_off, w.small_len = blob.end_group(w.norm1)
        w.qkv_w = blob.planes(blk.BMSA.qkv)
        w.proj_w = blob.planes(blk.BMSA.proj)
        w.w1_w = blob.planes(blk.swiglu.w1)
        w.w2_w = blob.planes(blk.swiglu.w2)
        w.w3_w = blob.planes(blk.swiglu.w3)


def _fill_resample(dst, mod, blob, train):
    dst.stride = mod.stride
    bn = mod.norm
    dst.bn_eps = float(bn.eps)
    dst.bn_momentum = float(bn.momentum) if bn.momentum is not None else 0.0
    dst.b = blob.add(mod.proj.bias.detach().float())
    if train:  # batch statistics: the kernel needs gamma / beta themselves (haet_erwin_train carries the running stats)
        dst.bn_scale = blob.add(bn.weight.detach().float())
        dst.bn_shift = blob.add(bn.bias.detach().float())
    else:
        scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
        dst.bn_scale = blob.add(scale)
        dst.bn_shift = blob.add(bn.bias.detach().float() - bn.running_mean.float() * scale)
    dst.small_off, dst.small_len = blob.end_group(dst.b)
    dst.w = blob.planes(mod.proj)


def pack_erwin(erwin):
    n_levels = len(erwin.encoder) + 1
    if n_levels > MAX_LEVELS:
        raise HaetError(f"Erwin with {n_levels} levels > {MAX_LEVELS}")
    device = next(erwin.parameters()).device
    desc, blob = ErwinDesc(), _Blob(device)
    desc.n_levels, desc.d = n_levels, erwin.dimensionality
    desc.rotate, desc.decode = int(erwin.rotate > 0), int(bool(erwin.decode))
    train = bool(erwin.training)
    desc.bn_train = int(train)
    for i, layer in enumerate(erwin.encoder):
        _fill_layer(desc.enc[i], layer, blob)
        _fill_resample(desc.pool[i], layer.pool, blob, train)
    _fill_layer(desc.enc[n_levels - 1], erwin.bottleneck, blob)
    # decoder[k] runs at level n_levels-2-k (erwin_flash.py:108-123)
    for k, layer in enumerate(erwin.decoder):
        lvl = n_levels - 2 - k
        _fill_layer(desc.dec[lvl], layer, blob)
        _fill_resample(desc.unpool[lvl], layer.unpool, blob, train)
    return blob.tensor(desc), desc
