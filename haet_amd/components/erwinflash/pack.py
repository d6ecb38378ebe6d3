"""Pack an ErwinTransformer's parameters into the single blob + descriptor that
haet_erwin_fwd consumes (layout contract: include/haet_hip.h, haet_erwin_desc).

Blob = [ copy of the descriptor | tensors ], offsets in floats from the blob start.
* Linear weights: PyTorch [out][in] layout, rows zero-padded to Kp = roundup(in, 32), stored
  pre-split as two bf16 planes -- hi = RNE(w) at the offset, lo = RNE(w - hi) O*Kp bf16 later --
  so the kernel feeds them to the MFMA without conversion (3 products hi*hi + hi*lo + lo*hi
  give fp32-class accuracy); behind the two planes (2*O*Kp bf16 from the offset) a tile-ordered copy of both
  (roundup(O, 16)*Kp*2 bf16: `_Blob.planes`), which is what erwin.hip's Linear loads.
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
        """[out][in~32] bf16 hi plane followed by the lo plane (see module docstring), followed by the TILE-ORDERED copy
        erwin.hip's Linear streams (round 4): per (16-row tile ot, 32-wide k-step ks, plane) one 1-KB piece in which lane
        l = 16 kq + r of a wave holds W[16 ot + r][32 ks + 8 kq .. + 8] -- a wave-wide 16-B load is then 1 KB of consecutive
        bytes instead of 16 rows x 64 B (which the vector cache serves lane by lane: ~11 B/cycle per CU, stamped at config C2).
        Rows past `out` are zero."""
        w = linear.weight.detach().float()
        pad = (-w.shape[1]) % 32
        if pad:
            w = torch.nn.functional.pad(w, (0, pad))
        hi = w.to(torch.bfloat16)
        lo = (w - hi.float()).to(torch.bfloat16)
        O, Kp = w.shape
        nto, nks = (O + 15) // 16, Kp // 32
        both = torch.stack([hi, lo])                                            # [plane][O][Kp]
        if nto * 16 != O:
            both = torch.nn.functional.pad(both, (0, 0, 0, nto * 16 - O))
        tiled = both.reshape(2, nto, 16, nks, 4, 8).permute(1, 3, 0, 4, 2, 5)    # [ot][ks][plane][kq][r][8]
        planes = torch.cat([hi.reshape(-1), lo.reshape(-1), tiled.reshape(-1)]).contiguous()
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
    """The direct packer: the layout's definition and the tests' reference -- the modules pack through ErwinPackPlan below."""
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


# --------------------------------------------------------------------------- fp32 blob of the backward kernel
def pack_erwin_f32(erwin):
    """(blob, layout, param_map) for haet_erwin_bwd: every tensor in fp32, Linear weights in PyTorch's [out][in]
    layout AND transposed (the kernel reads whichever makes its loads coalesced), each of the block Linears' two tensors
    followed by a tile-ordered copy of itself (round 4: what the matrix-core phases load); the blob BEGINS WITH the layout
    (include/haet_hip.h haet_erwin_f32_layout).  `param_map`: [(parameter, offset)] -- where each parameter's gradient
    sits in a cloud's slice of the kernel's `grads` output."""
    from ..._lib import ErwinF32Layout

    n_levels = len(erwin.encoder) + 1
    device = next(erwin.parameters()).device
    lay = ErwinF32Layout()
    head = (ctypes.sizeof(ErwinF32Layout) + 15) // 16 * 4
    chunks, pmap = [], []
    state = {"n": head}

    def add(t, param=None):
        t = t.detach().float().reshape(-1)
        off = state["n"]
        chunks.append(t.to(device))
        state["n"] += t.numel()
        pad = (-state["n"]) % 4
        if pad:
            chunks.append(torch.zeros(pad, device=device))
            state["n"] += pad
        if param is not None:
            pmap.append((param, off))
        return off

    def tiled(w):
        """[R][K] -> per (32-row tile, 16-wide k-step) two 1-KB pieces in which lane (r, hf) of a wave holds
        W[32 ot + r][16 s + 8 hf + 4 which .. + 4] (erwin_bwd.hip eb_phase, wfetch); same number of floats; zeros where the
        shape is not a multiple of (32, 16) -- such shapes never take the matrix-core phases."""
        w = w.detach().float()
        R, K = w.shape
        if R % 32 or K % 16:
            return torch.zeros(R * K, device=w.device)
        # [ot][r][s][hf][which][4] -> [ot][s][which][hf][r][4]
        return w.reshape(R // 32, 32, K // 16, 2, 2, 4).permute(0, 2, 4, 3, 1, 5).contiguous().reshape(-1)

    def fill_block(dst, blk):
        a = blk.BMSA
        dst.norm1, dst.norm2 = add(blk.norm1.weight, blk.norm1.weight), add(blk.norm2.weight, blk.norm2.weight)
        dst.pe_w, dst.pe_b = add(a.pe_proj.weight, a.pe_proj.weight), add(a.pe_proj.bias, a.pe_proj.bias)
        for name, lin in (("qkv", a.qkv), ("proj", a.proj), ("w1", blk.swiglu.w1), ("w2", blk.swiglu.w2), ("w3", blk.swiglu.w3)):
            setattr(dst, name + "_w", add(lin.weight, lin.weight))
            add(tiled(lin.weight))      # (round 4) the tile-ordered copy the matrix-core phases stream: right behind the tensor
            setattr(dst, name + "_wT", add(lin.weight.detach().t().contiguous()))
            add(tiled(lin.weight.detach().t()))
            setattr(dst, name + "_b", add(lin.bias, lin.bias))

    def fill_resample(dst, mod):
        dst.w, dst.wT = add(mod.proj.weight, mod.proj.weight), add(mod.proj.weight.detach().t().contiguous())
        dst.b = add(mod.proj.bias, mod.proj.bias)
        dst.gamma, dst.beta = add(mod.norm.weight, mod.norm.weight), add(mod.norm.bias, mod.norm.bias)
        dst.rmean, dst.rvar = add(mod.norm.running_mean), add(mod.norm.running_var)

    for i, layer in enumerate(erwin.encoder):
        for j, blk in enumerate(layer.blocks):
            fill_block(lay.enc[i][j], blk)
        fill_resample(lay.pool[i], layer.pool)
    for j, blk in enumerate(erwin.bottleneck.blocks):
        fill_block(lay.enc[n_levels - 1][j], blk)
    for k, layer in enumerate(erwin.decoder):
        lvl = n_levels - 2 - k
        for j, blk in enumerate(layer.blocks):
            fill_block(lay.dec[lvl][j], blk)
        fill_resample(lay.unpool[lvl], layer.unpool)
    lay.total = state["n"]
    hdr = torch.frombuffer(bytearray(bytes(lay)) + bytearray(head * 4 - ctypes.sizeof(lay)), dtype=torch.float32).to(device)
    return torch.cat([hdr] + chunks).contiguous(), lay, pmap


# --------------------------------------------------------------------------- gather plans (round 4)
# pack_erwin / pack_erwin_f32 above are ~12 torch ops per Linear: ~400 launches per ErwinTransformer and repack.  At
# inference the blob is built once; in TRAINING every optimizer step changes the weights, so every step repacks -- at the
# reference's PDE-benchmark widths (8 layers, n_hidden 128, 1k-17k points: benchmarks/04-.../scripts/*.sh) that was 2/3 of
# the ~4 000 launches of a step, and the step was host-bound at 66-110 ms.  Both blobs are pure rearrangements of the
# parameters (plus the bf16 hi / lo split), so the layout is computed ONCE as an index tensor -- by the same sequence of
# reshape / pad / permute / cat, applied to the elements' source INDICES instead of their values -- and a repack is
#     flat = cat(parameters);  hi = bf16(flat);  lo = bf16(flat - hi);  blob = cat(flat | hi | lo | header)[plan]
# i.e. ~8 launches whatever the depth.  tests/test_modules_cpu.py holds the plans to the direct packers bit for bit.
class _Sources:
    """The live tensors the blobs are made of, in first-use order; `sym(t)` = t's element indices in `flat()` =
    [0.0 | sources ...]: element 0 is the zero every padding element gathers, so that plans sharing one registry stay valid
    when another plan registers more sources behind theirs."""

    def __init__(self):
        self.items, self.base, self.n = [], {}, 1

    def sym(self, t):
        k = id(t)
        if k not in self.base:
            self.base[k] = self.n
            self.items.append(t)
            self.n += t.numel()
        return torch.arange(self.base[k], self.base[k] + t.numel(), dtype=torch.int64).reshape(t.shape)

    def computed(self, fn, numel):
        """A derived vector (eval-mode BatchNorm folded to scale / shift): `fn()` is evaluated at every repack."""
        off = self.n
        self.items.append(fn)
        self.n += numel
        return torch.arange(off, off + numel, dtype=torch.int64)

    def flat(self, device):
        """[0.0 | fp32 concatenation of the sources].  The flattened fp32 views of the live tensors are kept between calls
        (an optimizer step writes into the same storage); a view is re-made when its tensor's storage moved (`p.data = ...`)."""
        views = self.__dict__.get("_views")
        if views is None or views[0] != str(device) or len(views[1]) != len(self.items) + 1:
            views = (str(device), [torch.zeros(1, device=device)] + [None] * len(self.items), [0] * len(self.items))
            self._views = views
        _, vs, ptrs = views
        for i, t in enumerate(self.items):
            if callable(t):
                vs[i + 1] = t().reshape(-1).to(device=device, dtype=torch.float32)
            elif vs[i + 1] is None or ptrs[i] != t.data_ptr():
                v = t.detach().reshape(-1)
                direct = v.device == vs[0].device and v.dtype == torch.float32 and t.is_contiguous()  # (else reshape / to made a copy)
                vs[i + 1] = v if direct else v.to(device=device, dtype=torch.float32)
                ptrs[i] = t.data_ptr() if direct else -1  # (a converted copy is re-made every time)
        return torch.cat(vs)


class _SymBlob:
    """_Blob on indices: parts are ('f32', idx) = fp32 words or ('b16', enc) = bf16 elements with enc = 2 * source + plane
    (plane 0 = hi, 1 = lo); index -1 = zero."""

    def __init__(self):
        self.parts, self.n = [], HEADER_FLOATS

    def _pad(self):
        pad = (-self.n) % 4
        if pad:
            self.parts.append(("f32", torch.full((pad,), -1, dtype=torch.int64)))
            self.n += pad

    def add(self, sym):
        sym = sym.reshape(-1)
        self._pad()
        off = self.n
        self.parts.append(("f32", sym))
        self.n += sym.numel()
        return off

    def planes(self, w):
        pad = (-w.shape[1]) % 32
        if pad:
            w = torch.nn.functional.pad(w, (0, pad), value=-1)
        hi = torch.where(w >= 0, 2 * w, w)
        lo = torch.where(w >= 0, 2 * w + 1, w)
        O, Kp = w.shape
        nto, nks = (O + 15) // 16, Kp // 32
        both = torch.stack([hi, lo])
        if nto * 16 != O:
            both = torch.nn.functional.pad(both, (0, 0, 0, nto * 16 - O), value=-1)
        tiled = both.reshape(2, nto, 16, nks, 4, 8).permute(1, 3, 0, 4, 2, 5)
        enc = torch.cat([hi.reshape(-1), lo.reshape(-1), tiled.reshape(-1)])
        self._pad()
        off = self.n
        self.parts.append(("b16", enc))
        self.n += enc.numel() // 2
        return off

    def end_group(self, start):
        self._pad()
        return start, self.n - start

    def index16(self):
        """Per 16-bit unit of the blob (header excluded): (part, element) with part 0 = the fp32 word's low half, 1 = its
        high half, 2 = bf16 hi, 3 = bf16 lo of flat[element]; element 0 is the zero.  Resolved against the flat vector's
        length at run time (ErwinPackPlan.run), which may grow when another plan registers sources in the same registry."""
        self._pad()
        part, elem = [], []
        for kind, v in self.parts:
            s = torch.where(v >= 0, v if kind == "f32" else v >> 1, torch.zeros_like(v))
            if kind == "f32":
                elem.append(torch.stack([s, s], dim=1).reshape(-1))
                part.append(torch.tensor([0, 1], dtype=torch.int64).repeat(v.numel()))
            else:
                elem.append(s)
                part.append(torch.where(v >= 0, 2 + (v & 1), torch.full_like(v, 2)))
        return torch.cat(part), torch.cat(elem)


def _sym_layer(dst, layer, blob, src):
    dst.c, dst.heads, dst.ball = layer.hidden_dim, layer.num_heads, layer.ball_size
    dst.depth, dst.hidden = len(layer.blocks), layer.hidden_dim * layer.mlp_ratio
    if dst.depth > MAX_DEPTH:
        raise HaetError(f"Erwin layer depth {dst.depth} > {MAX_DEPTH}")
    for j, blk in enumerate(layer.blocks):
        w = dst.blocks[j]
        w.norm1 = blob.add(src.sym(blk.norm1.weight))
        w.norm2 = blob.add(src.sym(blk.norm2.weight))
        w.pe_w = blob.add(src.sym(blk.BMSA.pe_proj.weight).t())
        w.pe_b = blob.add(src.sym(blk.BMSA.pe_proj.bias))
        w.qkv_b = blob.add(src.sym(blk.BMSA.qkv.bias))
        w.proj_b = blob.add(src.sym(blk.BMSA.proj.bias))
        w.w1_b = blob.add(src.sym(blk.swiglu.w1.bias))
        w.w2_b = blob.add(src.sym(blk.swiglu.w2.bias))
        w.w3_b = blob.add(src.sym(blk.swiglu.w3.bias))
        w.small_off, w.small_len = blob.end_group(w.norm1)
        w.qkv_w = blob.planes(src.sym(blk.BMSA.qkv.weight))
        w.proj_w = blob.planes(src.sym(blk.BMSA.proj.weight))
        w.w1_w = blob.planes(src.sym(blk.swiglu.w1.weight))
        w.w2_w = blob.planes(src.sym(blk.swiglu.w2.weight))
        w.w3_w = blob.planes(src.sym(blk.swiglu.w3.weight))


def _sym_resample(dst, mod, blob, src, train):
    dst.stride = mod.stride
    bn = mod.norm
    dst.bn_eps = float(bn.eps)
    dst.bn_momentum = float(bn.momentum) if bn.momentum is not None else 0.0
    dst.b = blob.add(src.sym(mod.proj.bias))
    if train:
        dst.bn_scale = blob.add(src.sym(bn.weight))
        dst.bn_shift = blob.add(src.sym(bn.bias))
    else:  # (the same expressions as _fill_resample, evaluated at every repack)
        def scale():
            return bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)

        def shift():
            return bn.bias.detach().float() - bn.running_mean.float() * scale()

        dst.bn_scale = blob.add(src.computed(scale, bn.weight.numel()))
        dst.bn_shift = blob.add(src.computed(shift, bn.weight.numel()))
    dst.small_off, dst.small_len = blob.end_group(dst.b)
    dst.w = blob.planes(src.sym(mod.proj.weight))


class ErwinPackPlan:
    """Layout of pack_erwin's blob (for the module's current mode) as a gather; `run(flat)` = the blob of the current weights.
    The descriptor is part of the plan: it holds offsets and hyper-parameters only, so it is the same object for every repack.
    `src`: the source registry, shared with the module's other plans so that one `flat()` serves them all."""

    def __init__(self, erwin, src=None):
        n_levels = len(erwin.encoder) + 1
        if n_levels > MAX_LEVELS:
            raise HaetError(f"Erwin with {n_levels} levels > {MAX_LEVELS}")
        self.device = next(erwin.parameters()).device
        desc, blob, src = ErwinDesc(), _SymBlob(), (_Sources() if src is None else src)
        desc.n_levels, desc.d = n_levels, erwin.dimensionality
        desc.rotate, desc.decode = int(erwin.rotate > 0), int(bool(erwin.decode))
        train = bool(erwin.training)
        desc.bn_train = int(train)
        for i, layer in enumerate(erwin.encoder):
            _sym_layer(desc.enc[i], layer, blob, src)
            _sym_resample(desc.pool[i], layer.pool, blob, src, train)
        _sym_layer(desc.enc[n_levels - 1], erwin.bottleneck, blob, src)
        for k, layer in enumerate(erwin.decoder):
            lvl = n_levels - 2 - k
            _sym_layer(desc.dec[lvl], layer, blob, src)
            _sym_resample(desc.unpool[lvl], layer.unpool, blob, src, train)
        self.desc, self.src = desc, src
        part, elem = blob.index16()
        self.part, self.elem = part.to(torch.int8), elem.to(torch.int32)  # (host copies: 5 B per 16-bit unit of the blob)
        self._index = None  # (n, device index): resolved for a flat vector of n elements, 4 B per unit
        hdr = bytearray(bytes(desc)) + bytearray(HEADER_FLOATS * 4 - ctypes.sizeof(desc))
        self.hdr16 = torch.frombuffer(hdr, dtype=torch.int16).to(self.device)

    def index(self, n):
        """Indices into  [flat as int16 (2 n) | hi (n) | lo (n) | header]  for a flat vector of n elements."""
        if self._index is None or self._index[0] != n:
            p, e = self.part.to(torch.int32), self.elem
            body = torch.where(p < 2, 2 * e + p, p * n + e)  # (hi at 2 n + e, lo at 3 n + e)
            hdr = 4 * n + torch.arange(self.hdr16.numel(), dtype=torch.int32)
            self._index = (n, torch.cat([hdr, body]).to(self.device))
        return self._index[1]

    def run(self, flat=None):
        if flat is None:
            flat = self.src.flat(self.device)
        hi = flat.to(torch.bfloat16)
        lo = (flat - hi.float()).to(torch.bfloat16)
        pool = torch.cat([flat.view(torch.int16), hi.view(torch.int16), lo.view(torch.int16), self.hdr16])
        return torch.index_select(pool, 0, self.index(flat.numel())).view(torch.float32)


class ErwinF32PackPlan:
    """Layout of pack_erwin_f32's blob as a gather of fp32 words; `.lay`, `.pmap` as pack_erwin_f32 returns them."""

    def __init__(self, erwin, src=None):
        from ..._lib import ErwinF32Layout

        n_levels = len(erwin.encoder) + 1
        self.device = next(erwin.parameters()).device
        lay, src = ErwinF32Layout(), (_Sources() if src is None else src)
        head = (ctypes.sizeof(ErwinF32Layout) + 15) // 16 * 4
        parts, pmap = [], []
        state = {"n": head}

        def add(sym, param=None):
            sym = sym.reshape(-1)
            off = state["n"]
            parts.append(sym)
            state["n"] += sym.numel()
            pad = (-state["n"]) % 4
            if pad:
                parts.append(torch.full((pad,), -1, dtype=torch.int64))
                state["n"] += pad
            if param is not None:
                pmap.append((param, off))
            return off

        def tiled(w):
            R, K = w.shape
            if R % 32 or K % 16:
                return torch.full((R * K,), -1, dtype=torch.int64)
            return w.reshape(R // 32, 32, K // 16, 2, 2, 4).permute(0, 2, 4, 3, 1, 5).reshape(-1)

        def fill_block(dst, blk):
            a = blk.BMSA
            dst.norm1, dst.norm2 = add(src.sym(blk.norm1.weight), blk.norm1.weight), add(src.sym(blk.norm2.weight), blk.norm2.weight)
            dst.pe_w, dst.pe_b = add(src.sym(a.pe_proj.weight), a.pe_proj.weight), add(src.sym(a.pe_proj.bias), a.pe_proj.bias)
            for name, lin in (("qkv", a.qkv), ("proj", a.proj), ("w1", blk.swiglu.w1), ("w2", blk.swiglu.w2), ("w3", blk.swiglu.w3)):
                w = src.sym(lin.weight)
                setattr(dst, name + "_w", add(w, lin.weight))
                add(tiled(w))
                setattr(dst, name + "_wT", add(w.t()))
                add(tiled(w.t()))
                setattr(dst, name + "_b", add(src.sym(lin.bias), lin.bias))

        def fill_resample(dst, mod):
            w = src.sym(mod.proj.weight)
            dst.w, dst.wT = add(w, mod.proj.weight), add(w.t())
            dst.b = add(src.sym(mod.proj.bias), mod.proj.bias)
            dst.gamma, dst.beta = add(src.sym(mod.norm.weight), mod.norm.weight), add(src.sym(mod.norm.bias), mod.norm.bias)
            dst.rmean, dst.rvar = add(src.sym(mod.norm.running_mean)), add(src.sym(mod.norm.running_var))

        for i, layer in enumerate(erwin.encoder):
            for j, blk in enumerate(layer.blocks):
                fill_block(lay.enc[i][j], blk)
            fill_resample(lay.pool[i], layer.pool)
        for j, blk in enumerate(erwin.bottleneck.blocks):
            fill_block(lay.enc[n_levels - 1][j], blk)
        for k, layer in enumerate(erwin.decoder):
            lvl = n_levels - 2 - k
            for j, blk in enumerate(layer.blocks):
                fill_block(lay.dec[lvl][j], blk)
            fill_resample(lay.unpool[lvl], layer.unpool)
        lay.total = state["n"]
        self.lay, self.pmap, self.src = lay, pmap, src
        hdr = bytearray(bytes(lay)) + bytearray(head * 4 - ctypes.sizeof(lay))
        self.hdr = torch.frombuffer(hdr, dtype=torch.float32).to(self.device)
        idx = torch.cat(parts)
        self.body = torch.where(idx >= 0, idx, torch.zeros_like(idx)).to(torch.int32)  # (host copy; element 0 of flat = the zero)
        self._index = None

    def index(self, n):
        if self._index is None or self._index[0] != n:
            hdr = n + torch.arange(self.hdr.numel(), dtype=torch.int32)
            self._index = (n, torch.cat([hdr, self.body]).to(self.device))
        return self._index[1]

    def run(self, flat=None):
        if flat is None:
            flat = self.src.flat(self.device)
        return torch.index_select(torch.cat([flat, self.hdr]), 0, self.index(flat.numel()))
