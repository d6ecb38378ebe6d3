"""Shared implementation of the three Physics_Attention_* modules.

The reference copy-pastes one forward three times (models/PhysicsAttention/
IrregularMesh.py:141-235, StructuredMesh2D.py:157-262, StructuredMesh3D.py:161-266); the
variants differ only in the per-point projection (Linear / Conv2d / Conv3d).  Every stage is a
hand-written HIP kernel behind the C ABI (rocBLAS / MIOpen only for channel counts outside the
GEMM / implicit-GEMM kernels' envelopes):

    x --haet_project_split_fwd | haet_conv_project_fwd--> x_lin [B,N,H*C] (fused operand format or fp32)
      --haet_slice_fwd[_split]--> partial (m,l,St) --haet_slice_merge--> S, col stats
      --haet_token_stage_fwd (positions + ball tree + Erwin, one launch)--> S'
      --haet_deslice_out_fwd_split (de-slice + to_out fused) | haet_deslice_fwd + haet_linear_fwd--> [B,N,dim]

The [B,H,N,G] slice-weight tensor of the reference (2 GB at N=1M) is never materialised.
With `shard_group` set, each rank holds a shard of the points and the partial records are
all-gathered and merged (one tiny exchange per forward); everything after the
merge is replicated.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Optional

import torch
import torch.nn as nn

from .. import _compile
from .. import _lib as _lib_mod
from .. import ops
from ..components import ErwinFlashTransformer as ErwinTransformer
from .._lib import HaetError


class _StageTimer:
    """Brackets a stage with HIP events on the current (= launch) stream when enabled.  Events come
    from `pool` when the caller provides one (the first record() of a fresh timing event creates the
    hipEvent: ~60 us of host time).  Consecutive stages share the event between them (`chain[0]`, the
    end of the previous stage): a recorded event costs ~3-5 us on the GPU timeline as well, and
    fourteen of them were 10 % of a 125k-point forward."""

    def __init__(self, sink, name, pool=None, chain=None):
        self.sink, self.name, self.pool, self.chain = sink, name, pool, chain

    def _event(self):
        e = self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def __enter__(self):
        if self.sink is not None:
            prev = self.chain[0] if self.chain is not None else None
            self.e0 = prev if prev is not None else self._event()

    def __exit__(self, *exc):
        if self.sink is not None:
            e1 = self._event()
            if self.chain is not None:
                self.chain[0] = e1
            self.sink.append((self.name, self.e0, e1))
        return False


def _invalidate_hook(module, incompatible):
    module.invalidate_caches()


# Live attention modules: the Ada-Temp bias is a SCALAR kernel argument, i.e. a device -> host read whenever the weights
# changed -- once per layer and optimizer step if every module reads its own, and each read drains the launch queue (at
# the reference's PDE-benchmark sizes a step is launch-bound: the host has to run ahead of the GPU).  The first module
# that finds its parameters changed reads the scalars of ALL live modules on its device in one copy.
_LIVE = weakref.WeakSet()


def _refresh_slice_params(first):
    dev = first.ada_temp_linear.bias.device
    _LIVE.add(first)  # (modules made by copy / unpickling never ran __init__)
    stale = []
    for m in [first] + [m for m in list(_LIVE) if m is not first]:
        b = m.ada_temp_linear.bias
        if b.device != dev or b.numel() != 1:
            continue
        key = m._slice_key()
        if m is first or m._slice_cache is None or m._slice_cache[0] != key:
            stale.append((m, key))
    # (these copies synchronise: the point at which a timed-out grid barrier of the previous training step's Erwin
    # launches is reported -- one read of all the modules' barrier words)
    ops.check_barriers([st for st in (m.erwin.train_state() for m, _ in stale) if st is not None])
    bts = torch.cat([m.ada_temp_linear.bias.detach().reshape(1).float() for m, _ in stale]).tolist()
    for (m, key), bt in zip(stale, bts):
        ps = m._slice_tensors()
        m._slice_cache = (key, ops.SliceParams(ps[0].detach(), ps[1].detach(), ps[2].detach(), float(bt), m.base_temp, m.epsilon))


class PhysicsAttentionBase(_compile.OpaqueForwardMixin, nn.Module):
    dimensionality = 3

    def _init_common(self, dim, heads, dim_head, dropout, slice_num, base_temp, epsilon, c_hidden,
                     ball_sizes, enc_num_heads, enc_depths, dec_num_heads, dec_depths, strides,
                     rotate, decode, mlp_ratio, mp_steps, embed):
        inner_dim = dim_head * heads
        self.dim_head, self.heads, self.slice_num = dim_head, heads, slice_num
        self.epsilon, self.base_temp = epsilon, base_temp
        self.in_project_slice = nn.Linear(dim_head, slice_num)
        self.ada_temp_linear = nn.Linear(dim_head, 1)
        torch.nn.init.orthogonal_(self.in_project_slice.weight)
        # defaults of IrregularMesh.py:105-118
        if c_hidden is None:
            c_hidden = [dim_head, dim_head * 2]
        if ball_sizes is None:
            ball_sizes = [min(32, slice_num), min(16, slice_num // 2)]
        if enc_num_heads is None:
            enc_num_heads = [heads // 2, heads]
        if enc_depths is None:
            enc_depths = [2, 2]
        if dec_num_heads is None:
            dec_num_heads = [heads // 2]
        if dec_depths is None:
            dec_depths = [2]
        if strides is None:
            strides = [2]
        if embed or mp_steps:
            raise NotImplementedError(
                "ErwinEmbedding / MPNN inside PhysicsAttention (embed=True or mp_steps>0): the slice-token stage is one fused "
                "launch without an embedding -- every Physics_Attention_* default passes embed=False, mp_steps=0, and the "
                "reference's own forward gives the MPNN no radius / edge_index (IrregularMesh.py:211-218).  The embedding "
                "exists on the stand-alone ErwinTransformer (haet_amd.components.erwinflash).")
        self.erwin = ErwinTransformer(
            c_in=dim_head, c_hidden=c_hidden, ball_sizes=ball_sizes, enc_num_heads=enc_num_heads,
            enc_depths=enc_depths, dec_num_heads=dec_num_heads, dec_depths=dec_depths, strides=strides,
            rotate=rotate, decode=decode, mlp_ratio=mlp_ratio, dimensionality=self.dimensionality,
            mp_steps=mp_steps, embed=embed)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))
        self.shard_group = None  # torch.distributed group over which the POINTS are sharded
        self.stage_events = None  # set to [] to collect (stage, start, end) HIP events (bench.py)
        self.event_pool = None    # optional list of pre-recorded timing events the stage timer draws from
        self._slice_cache = None
        _LIVE.add(self)
        _compile.register(self)
        self.register_load_state_dict_post_hook(_invalidate_hook)  # (a module-level function: the module stays picklable)

    # -- per-variant projection to [B, N, H*C] -------------------------------------------
    def project(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def project_split(self, x: torch.Tensor, sp, pre_norm=None):
        """Projection straight into the streaming passes' operand format (ops.SplitPoints), or None
        when this variant / shape has no such path (then `project` + the fp32 passes run).  `pre_norm`:
        a LayerNorm to apply to x inside the projection (only ever passed when the variant supports it)."""
        return None

    def fuses_pre_norm(self, x: torch.Tensor, ln) -> bool:
        """Whether `project_split(x, sp, pre_norm=ln)` can take the block's LayerNorm into its loader."""
        return False

    def _conv_fuses_pre_norm(self, x: torch.Tensor, ln) -> bool:
        """The block's LayerNorm inside the pre-split pass of the LDS-DMA convolution (haet_conv_presplit): C_in = 256, the
        shapes that kernel serves, the split operand format downstream."""
        conv = self.in_project_x
        B, N, _ = x.shape
        return (ln.elementwise_affine and ln.bias is not None and tuple(ln.normalized_shape) == (conv.in_channels,)
                and conv.in_channels == 256 and x.dtype == torch.float32 and not os.environ.get("HAET_NO_HIP_CONV")
                and not os.environ.get("HAET_NO_SPLIT_FORMAT") and self.dim_head == 32 and self._conv_is_plain(conv)
                and ops.conv_cells_applies(conv.in_channels, conv.out_channels, B * N, math.prod(conv.kernel_size))
                and ops.slice_split_supported(B, N, self.heads, self.dim_head, self.slice_num))

    @staticmethod
    def _conv_is_plain(conv) -> bool:
        return not (conv.bias is None or any(s != 1 for s in conv.stride) or any(d != 1 for d in conv.dilation)
                    or conv.groups != 1 or any(k % 2 == 0 or p != k // 2 for k, p in zip(conv.kernel_size, conv.padding))
                    or conv.padding_mode != "zeros")

    def _conv_project_split(self, x: torch.Tensor, grid, sp, pre_norm=None):
        """The Conv2d / Conv3d projection with Ada-Temp in its epilogue and the output in the fused operand format
        (ops.SplitPoints), or None where that path does not apply (dim_head != 32, channel counts outside the HIP
        convolution, a shape the split-format passes do not take)."""
        B, N, _ = x.shape
        if (os.environ.get("HAET_NO_SPLIT_FORMAT") or self.dim_head != 32
                or not ops.slice_split_supported(B, N, self.heads, self.dim_head, self.slice_num)):
            return None
        return self._conv_project(x, grid, split=sp, pre_norm=pre_norm)

    def _conv_project(self, x: torch.Tensor, grid, split=None, pre_norm=None):
        """Conv2d / Conv3d in_project_x as the HIP implicit GEMM over the channels-last points, or None
        when the channel counts are outside its envelope (the caller then runs the MIOpen conv).  `split`: the
        slice parameters -> the result is an ops.SplitPoints (cells + s) instead of fp32 rows."""
        conv = self.in_project_x
        ok = not (os.environ.get("HAET_NO_HIP_CONV") or x.dtype != torch.float32)
        if ok and (not ops.conv_project_supported(conv.in_channels, conv.out_channels) or not self._conv_is_plain(conv)):
            ok = False
        if not ok:
            if self.shard_group is not None:
                raise HaetError("point-sharded structured meshes need the HIP convolution (fp32 input, C_in % 32 = 0, "
                                "heads * dim_head % 128 = 0, odd kernel with padding k // 2)")
            return None
        cache = self.__dict__.setdefault("_split_cache", {})
        key = (conv.weight.data_ptr(), conv.weight._version, conv.bias.data_ptr(), conv.bias._version)
        if "in_project_x" not in cache or cache["in_project_x"][0] != key:
            cache["in_project_x"] = (key, ops.SplitConv(conv.weight, conv.bias))
        w = cache["in_project_x"][1]
        ln = None if pre_norm is None else (pre_norm.weight, pre_norm.bias, pre_norm.eps)  # (only where _conv_fuses_pre_norm)
        run = ((lambda xx, gg: ops.conv_project(xx, w, gg, ln)) if split is None
               else (lambda xx, gg: ops.conv_project_split(xx, w, gg, split, self.heads, self.dim_head, ln)))
        if self.shard_group is None:
            return run(x, grid)
        # Slab sharding along the first grid axis: this rank holds whole planes; the convolution needs
        # one plane of halo per side per kernel radius (kernel 3: one plane), fetched from the
        # neighbour ranks; the global faces keep the zero padding.
        from ..sharding import exchange_halo_planes

        plane = 1
        for g in grid[1:]:
            plane *= g
        B, n, _ = x.shape
        if n % plane:
            raise HaetError("structured sharding: a rank must hold whole planes of the grid's first axis")
        rad = conv.kernel_size[0] // 2
        if rad > 1 or n // plane < rad:
            raise HaetError("structured sharding supports kernel 3 (one halo plane) and >= 1 plane per rank")
        prev, nxt = exchange_halo_planes(x, plane, self.shard_group) if rad else (None, None)
        parts = [t for t in (prev, x, nxt) if t is not None]
        x_ext = torch.cat(parts, dim=1) if len(parts) > 1 else x
        y = run(x_ext, (x_ext.shape[1] // plane,) + tuple(grid[1:]))
        lo = plane if prev is not None else 0
        if split is not None:  # the halo planes' rows are dropped from cells and s alike
            return ops.SplitPoints(y.cells[:, lo:lo + n].contiguous(), y.s[:, lo:lo + n].contiguous(), y.H, y.C)
        return y[:, lo:lo + n].contiguous()

    def _split_linear_applies(self, lin: nn.Linear, x: torch.Tensor) -> bool:
        B, N, _ = x.shape
        if os.environ.get("HAET_NO_SPLIT_FORMAT") or x.dtype != torch.float32:
            return False
        return (ops.project_split_supported(lin.in_features, lin.out_features, B * N)
                and ops.slice_split_supported(B, N, self.heads, self.dim_head, self.slice_num)
                and ops.linear_supported(lin.in_features, lin.out_features))

    def _project_split_linear(self, name: str, lin: nn.Linear, x: torch.Tensor, sp, pre_norm=None):
        if not self._split_linear_applies(lin, x):
            return None
        w = self._split_linear(name, lin)
        ln = None if pre_norm is None else (pre_norm.weight, pre_norm.bias, pre_norm.eps)
        return None if w is None else ops.project_split(x, w, sp, self.heads, self.dim_head, ln)

    def _split_linear(self, name: str, lin: nn.Linear):
        """haet_linear_fwd operand of a Linear (cached per weight version), or None when the shape
        is outside the MFMA GEMM's envelope (then the stock rocBLAS GEMM runs)."""
        if not ops.linear_supported(lin.in_features, lin.out_features):
            return None
        cache = self.__dict__.setdefault("_split_cache", {})
        key = (lin.weight.data_ptr(), lin.weight._version, lin.bias.data_ptr(), lin.bias._version)
        if name not in cache or cache[name][0] != key:
            cache[name] = (key, ops.SplitLinear(lin.weight, lin.bias))
        return cache[name][1]

    def _linear(self, name: str, lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
        w = self._split_linear(name, lin)
        return lin(x) if w is None else ops.linear(x, w)

    # -- dim_head 16 on the matrix-core passes ---------------------------------------------------------------------
    HEAD_PAD = 32
    HEAD_PAD_BWD_ROWS = 4096  # B * N from which the backward runs padded too (45k points, M = 64: 43 + 110 us against the
    #                           vector-ALU kernels' 97 + 253; below a few thousand rows everything is launch latency)

    def head_pad_applies(self, B: int, N: int) -> bool:
        """dim_head 16 (n_hidden 128 / 8 heads: six of the reference's seven PDE scripts) has no matrix-core slice /
        de-slice kernels of its own -- the generic vector-ALU ones ran there, 3-5x slower per byte.  The passes are run
        at 32 channels per head instead, the upper 16 zero: x_lin padded once per forward, W_s and w_t zero-padded
        (their products with the zero channels vanish exactly: same logits, same tau, same sums), the tokens cut back to
        16 channels around the Erwin stage and the output around `to_out`.  Two copies of x_lin-sized tensors per
        direction against kernels that run 3-4x faster; HAET_NO_HEAD_PAD=1 keeps the vector-ALU kernels (A/B, tests)."""
        return (self.dim_head == 16 and not os.environ.get("HAET_NO_HEAD_PAD")
                and ops.slice_mfma_supported(B, N, self.heads, self.HEAD_PAD, self.slice_num))

    def _slice_params_padded(self) -> ops.SliceParams:
        sp = self._slice_params()
        hit = sp.__dict__.get("_padded")
        if hit is None:
            extra = self.HEAD_PAD - self.dim_head
            hit = ops.SliceParams(torch.nn.functional.pad(sp.Ws, (0, extra)), sp.bs, torch.nn.functional.pad(sp.wt, (0, extra)),
                                  sp.bt, sp.base_temp, self.epsilon)
            sp.__dict__["_padded"] = hit
        return hit

    def _slice_tensors(self):
        return (self.in_project_slice.weight, self.in_project_slice.bias, self.ada_temp_linear.weight, self.ada_temp_linear.bias)

    def _slice_key(self):
        return tuple((p.data_ptr(), p._version) for p in self._slice_tensors())

    def _slice_params(self) -> ops.SliceParams:
        if self._slice_cache is None or self._slice_cache[0] != self._slice_key():
            _refresh_slice_params(self)
        return self._slice_cache[1]

    def _stage(self, name):
        return _StageTimer(self.stage_events, name, self.event_pool, self.__dict__.get("_event_chain"))

    def _any_requires_grad(self) -> bool:
        d = self.__dict__
        ps = d.get("_grad_check_params")
        gen = _lib_mod.cache_generation()
        if ps is None or d.get("_grad_check_stale", True) or d.get("_grad_check_gen") != gen:
            ps = list(self.parameters())  # ~50 us: cached, see ErwinTransformer._pack_key
            d["_grad_check_params"], d["_grad_check_stale"], d["_grad_check_gen"] = ps, False, gen
        for p in ps:
            if p.requires_grad:
                return True
        return False

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda(): parameter objects may be replaced
        self.__dict__["_grad_check_stale"] = True
        return super()._apply(fn, *args, **kwargs)

    def __getstate__(self):
        # cached operands (bf16 planes, slice parameters) and timing plumbing are per-process: rebuilt on demand
        state = dict(self.__dict__)
        for k in ("_split_cache", "_grad_check_params", "_event_chain", "_haet_handle"):
            state.pop(k, None)
        state["_slice_cache"] = None
        state["stage_events"] = state["event_pool"] = None
        state["_grad_check_stale"] = True
        return state

    def invalidate_caches(self):
        """Drop every weight-derived operand this module caches (bf16 planes of the projections, the slice
        parameters, the packed Erwin blob); they are re-derived at the next forward.  The caches are keyed on
        each parameter's (storage pointer, `_version`), which ordinary updates (optimizer steps, `copy_`,
        load_state_dict) change -- but in-place writes through `.data` (an EMA swap `p.data.copy_(shadow)`,
        `p.data.clamp_()`) change neither: call this after such writes."""
        self.__dict__["_split_cache"] = {}
        self._slice_cache = None
        self.__dict__["_grad_check_stale"] = True
        self.erwin.invalidate_caches()

    def forward(self, x: torch.Tensor, return_intermediates: bool = False, pre_norm: Optional[nn.LayerNorm] = None,
                residual: bool = False):
        # Under torch.compile without gradients the module is ONE opaque operator of the graph (haet_amd/_compile.py);
        # everything else is the eager path below, which Dynamo does not trace.
        if _compile.use_operator(x) and not return_intermediates and pre_norm is None and not residual:
            return torch.ops.haet.module_forward(x, _compile.primed_handle(self._haet_handle), x.shape[-1])
        return self._forward_eager(x, return_intermediates, pre_norm, residual)

    def _opaque_forward(self, x: torch.Tensor) -> torch.Tensor:  # body of haet::module_forward
        return self._forward_eager(x)

    def _prime_points(self) -> int:
        """Points of the dummy cloud used to build the weight caches ahead of a compiled run (structured variants: the grid)."""
        return 256

    def _prime(self):
        p = next(self.parameters())
        if p.is_cuda:
            was = self.training
            self.eval()  # (no running-statistics update from a dummy input)
            try:
                lin = self.in_project_x
                d_in = lin.in_features if isinstance(lin, nn.Linear) else lin.in_channels
                self._forward_eager(torch.zeros(1, self._prime_points(), d_in, device=p.device))
            finally:
                self.train(was)

    @torch.compiler.disable  # ctypes calls into libhaet_hip.so: an opaque eager call for Dynamo
    def _forward_eager(self, x: torch.Tensor, return_intermediates: bool = False, pre_norm: Optional[nn.LayerNorm] = None,
                       residual: bool = False):
        """`pre_norm`, `residual` (not in the reference's signature; used by the block's inference fast path): a
        LayerNorm to apply to x first -- fused into the projection GEMM's loader where the variant supports it -- and
        whether to return `forward(x) + x` (the block's residual, added in the fused de-slice + to_out epilogue)."""
        if residual:
            if not (x.is_cuda and not torch.is_grad_enabled() and not return_intermediates):
                return self._forward_eager(x, return_intermediates, pre_norm) + x
            res_src = x
        if pre_norm is not None and not (x.is_cuda and not torch.is_grad_enabled() and not self.training
                                         and self.fuses_pre_norm(x, pre_norm)):
            from .. import _fast  # (the HIP LayerNorm where it applies: ~10 us against torch's ~45 at 45k x 128)

            x, pre_norm = (_fast.layer_norm(pre_norm, x) if x.is_cuda and x.dtype == torch.float32 else pre_norm(x)), None
        if not x.is_cuda:
            raise HaetError("Physics_Attention_* (haet_amd) runs on a ROCm device only; move the module "
                            "and its input with .cuda() -- there is no CPU fallback")
        wants_grad = torch.is_grad_enabled() and (x.requires_grad or self._any_requires_grad())
        if wants_grad:
            # One autograd node whose N-scaled work is on the HIP kernels in both directions; only the
            # Erwin stage (B*H*M tokens) differentiates through a torch restatement -- haet_amd/_autograd.py.
            from .. import _autograd

            if return_intermediates:
                raise NotImplementedError("return_intermediates is a no-grad / eval facility")
            if os.environ.get("HAET_TORCH_BACKWARD") and self.shard_group is None:  # the interim path, for A/B runs
                if self.training:
                    return _autograd.attention_forward(self, x)
                return _autograd.HipForwardTorchBackward.apply(self, x, *self.parameters())
            y = _autograd.PhysicsAttentionFunction.apply(self, x, *self.parameters())
            return self.to_out[1](y).to(x.dtype)
        # no gradient wanted (inference; validation passes under no_grad, also in .train() mode: the Erwin kernel
        # then runs BatchNorm on batch statistics and updates the running ones, like the reference)
        return self._forward_hip(x, return_intermediates, pre_norm, res_src if residual else None)

    def project_torch(self, x: torch.Tensor) -> torch.Tensor:
        """The projection as stock torch ops (differentiable path, haet_amd/_autograd.py)."""
        raise NotImplementedError

    def project_backward(self, x: torch.Tensor, g_lin: torch.Tensor, need_x: bool, wanted: set):
        """Backward of `project`: (dL/dx or None, {id(param): grad}).  Generic version through autograd over
        `project_torch` (the structured variants: MIOpen's convolution backward)."""
        params = [p for p in self.in_project_x.parameters() if id(p) in wanted]
        if not need_x and not params:
            return None, {}
        if self.shard_group is not None:
            return self._conv_backward_sharded(x, g_lin, need_x, params)
        gx = self._conv_input_grad(g_lin) if need_x else None  # HIP implicit GEMM with the transposed, flipped kernel
        via_torch = need_x and gx is None
        hip_w = self._conv_weight_grad(g_lin, x, self.grid_shape()) if params and not via_torch else None
        if hip_w is not None:
            return gx, {id(p): g for p, g in zip(params, hip_w(params))}
        if via_torch or params:
            with torch.enable_grad():
                xr = x.detach().requires_grad_(bool(via_torch))
                y = self.project_torch(xr)
            got = torch.autograd.grad(y, ([xr] if via_torch else []) + params, g_lin.to(y.dtype), allow_unused=True)
            if via_torch:
                gx = got[0]
            return gx, {id(p): g for p, g in zip(params, got[1 if via_torch else 0:])}
        return gx, {}

    def _conv_weight_grad(self, g_lin: torch.Tensor, x: torch.Tensor, grid):
        """Weight / bias gradient of the convolution projection as one shifted row-contraction per kernel tap on the
        f32 matrix cores (ops.conv_wgrad), or None outside its envelope (then MIOpen).  Returns a function mapping
        the wanted parameter list to their gradients."""
        conv = self.in_project_x
        if (not isinstance(conv, (nn.Conv2d, nn.Conv3d)) or os.environ.get("HAET_NO_HIP_CONV") or g_lin.dtype != torch.float32
                or not ops.conv_wgrad_supported(conv.in_channels, conv.out_channels)
                or any(s != 1 for s in conv.stride) or any(d != 1 for d in conv.dilation) or conv.groups != 1
                or any(k % 2 == 0 or p != k // 2 for k, p in zip(conv.kernel_size, conv.padding))
                or conv.padding_mode != "zeros"):
            return None
        dW, db = ops.conv_wgrad(g_lin, x, grid, conv.kernel_size)
        return lambda params: [dW if p is conv.weight else db for p in params]

    def _conv_input_grad(self, g_lin: torch.Tensor, grid=None):
        """dL/dx of the same-padded stride-1 convolution projection = the convolution of dL/dy with the kernel
        transposed (in <-> out channels) and flipped: the same implicit-GEMM kernel as the forward.  None when the
        channel counts are outside its envelope (then MIOpen's backward-data runs)."""
        conv = self.in_project_x
        if not isinstance(conv, (nn.Conv2d, nn.Conv3d)) or os.environ.get("HAET_NO_HIP_CONV") or g_lin.dtype != torch.float32:
            return None
        if (not ops.conv_project_supported(conv.out_channels, conv.in_channels)
                or any(s != 1 for s in conv.stride) or any(d != 1 for d in conv.dilation) or conv.groups != 1
                or any(k % 2 == 0 or p != k // 2 for k, p in zip(conv.kernel_size, conv.padding))
                or conv.padding_mode != "zeros"):
            return None
        cache = self.__dict__.setdefault("_split_cache", {})
        key = (conv.weight.data_ptr(), conv.weight._version)
        if "in_project_x.T" not in cache or cache["in_project_x.T"][0] != key:
            # kernel transposed (in <-> out) and flipped in every spatial axis = its taps in reverse order: one gather with a
            # kept index (Tensor.flip costs ~0.7 ms of host time per call, and this runs per layer and optimizer step)
            w = conv.weight.detach()
            taps = w[0, 0].numel()
            rev = cache.get("tap_reverse")
            if rev is None or rev.device != w.device or rev.numel() != taps:
                rev = cache["tap_reverse"] = torch.arange(taps - 1, -1, -1, device=w.device)
            wt = torch.index_select(w.reshape(w.shape[0], w.shape[1], taps), 2, rev).transpose(0, 1).reshape(
                (w.shape[1], w.shape[0]) + tuple(w.shape[2:])).contiguous()
            cache["in_project_x.T"] = (key, ops.SplitConv(wt, None))
        return ops.conv_project(g_lin, cache["in_project_x.T"][1], self.grid_shape() if grid is None else grid)

    def _conv_backward_sharded(self, x: torch.Tensor, g_lin: torch.Tensor, need_x: bool, params):
        """Backward of the slab-sharded convolution projection (`_conv_project` with halos): the convolution's
        backward (MIOpen) on the slab extended by the neighbours' halo planes, then the REVERSE halo exchange --
        the gradient that landed on a neighbour's plane goes home and is added there.  The weight / bias
        gradients are this rank's additive shares (sums over its own output points)."""
        from ..sharding import exchange_halo_planes, exchange_planes

        conv = self.in_project_x
        grid = self.grid_shape()
        plane = 1
        for g in grid[1:]:
            plane *= g
        B, n, Cin = x.shape
        prev, nxt = exchange_halo_planes(x, plane, self.shard_group)
        x_ext = torch.cat([t for t in (prev, x, nxt) if t is not None], dim=1)
        lo = plane if prev is not None else 0
        ext = x_ext.shape[1] // plane
        nd = len(grid)
        # input gradient over the extended slab: the HIP implicit GEMM where the channel counts allow (dL/dy is
        # zero on the halo planes, whose outputs belong to the neighbours), else MIOpen through autograd
        gxe = None
        if need_x:
            zeros = (lambda t: None if t is None else torch.zeros((B, plane, g_lin.shape[-1]), dtype=g_lin.dtype, device=g_lin.device))
            g_ext = torch.cat([t for t in (zeros(prev), g_lin, zeros(nxt)) if t is not None], dim=1)
            gxe = self._conv_input_grad(g_ext, (ext,) + tuple(grid[1:]))
        via_torch = need_x and gxe is None
        got = ()
        if params and not via_torch:
            zeros = (lambda t: None if t is None else torch.zeros((B, plane, g_lin.shape[-1]), dtype=g_lin.dtype, device=g_lin.device))
            g_ext = torch.cat([t for t in (zeros(prev), g_lin, zeros(nxt)) if t is not None], dim=1)
            hip_w = self._conv_weight_grad(g_ext, x_ext, (ext,) + tuple(grid[1:]))
            if hip_w is not None:
                got, params_done = tuple(hip_w(params)), True
            else:
                params_done = False
        else:
            params_done = False
        if via_torch or (params and not params_done):
            with torch.enable_grad(), torch.autocast("cuda", enabled=False):
                xe = x_ext.detach().float().requires_grad_(bool(via_torch))
                xg = xe.view((B, ext) + tuple(grid[1:]) + (Cin,)).permute(0, nd + 1, *range(1, nd + 1))
                y = conv(xg).permute(0, *range(2, nd + 2), 1).reshape(B, ext * plane, -1)[:, lo:lo + n]
            got = torch.autograd.grad(y, ([xe] if via_torch else []) + list(params), g_lin.to(y.dtype), allow_unused=True)
            if via_torch:
                gxe, got = got[0], got[1:]
        if not need_x:
            return None, {id(p): g for p, g in zip(params, got)}
        from_prev, from_next = exchange_planes(gxe[:, :plane] if prev is not None else None,
                                               gxe[:, lo + n:] if nxt is not None else None, self.shard_group)
        gx = gxe[:, lo:lo + n].clone()
        if from_prev is not None:
            gx[:, :plane] += from_prev
        if from_next is not None:
            gx[:, n - plane:] += from_next
        return gx, {id(p): g for p, g in zip(params, got)}

    def grid_shape(self):
        """(H, W[, D]) of a structured variant (the first axis is the sharded one)."""
        raise NotImplementedError

    def _linear_transposed(self, name: str, lin: nn.Linear, g: torch.Tensor) -> torch.Tensor:
        """g [..., O] @ W [O, K] -> [..., K]: the input gradient of a Linear, on the same MFMA GEMM."""
        if not ops.linear_supported(lin.out_features, lin.in_features) or g.dtype != torch.float32:
            return g.to(lin.weight.dtype) @ lin.weight.detach()
        cache = self.__dict__.setdefault("_split_cache", {})
        key = (lin.weight.data_ptr(), lin.weight._version)
        name = name + ".T"
        if name not in cache or cache[name][0] != key:
            cache[name] = (key, ops.SplitLinear(lin.weight.detach().t().contiguous(), None))
        return ops.linear(g, cache[name][1])

    def _forward_hip(self, x: torch.Tensor, return_intermediates: bool = False, pre_norm=None, residual=None):
        B, N, _ = x.shape
        H, C, G = self.heads, self.dim_head, self.slice_num
        self.__dict__["_event_chain"] = [None] if self.stage_events is not None else None
        ev = self._stage
        sp = self._slice_params()
        xs = None
        with ev("project"):
            xs = self.project_split(x, sp, pre_norm)                     # fused operand format, or None
            if xs is None and pre_norm is not None:
                from .. import _fast

                x = _fast.layer_norm(pre_norm, x) if x.dtype == torch.float32 else pre_norm(x)
            x_lin = None if xs is not None else self.project(x).float().contiguous()  # [B, N, H*C]
            padded = xs is None and self.head_pad_applies(B, N)
            x_real = x_lin
            if padded:  # the passes below see dim_head 32 (zero upper channels); tokens and out_x are cut back
                x_lin, sp, Cr, C = ops.pad_heads(x_lin, H, C, self.HEAD_PAD), self._slice_params_padded(), C, self.HEAD_PAD
        with ev("slice"):
            parts = (ops.slice_partials_split(xs, sp, G) if xs is not None
                     else ops.slice_partials(x_lin, sp, H, C, G))       # [P, record]
        with ev("merge"):
            if self.shard_group is not None:
                from ..sharding import allgather_records

                local = ops.merge_records(parts, B, H, C, G, finalize=False)
                parts = allgather_records(local, self.shard_group)      # [R, record]
            col_max, col_sum, S = ops.merge_records(parts, B, H, C, G, finalize=True)
            if padded:
                S = S[:, :Cr].contiguous()
        with ev("erwin"):  # token positions are derived inside the tree launch (haet_tokens_tree_build)
            self.erwin.__dict__["keep_positions"] = bool(return_intermediates)
            S2 = self.erwin(S, None, n_clouds=B * H)                     # [B*H*G, C]
            pos = self.erwin.__dict__.get("last_positions")              # [B*H*G, d] (kept on request only)
            S2p = torch.nn.functional.pad(S2, (0, C - Cr)) if padded else S2
        w_out = None
        if xs is not None and not return_intermediates and ops.deslice_out_supported(H, C, G, self.to_out[0].out_features):
            w_out = self._split_linear("to_out", self.to_out[0])
        if w_out is not None:  # de-slice + to_out in one pass: out_x [B, N, H*C] never exists
            drop = self.training and self.to_out[1].p > 0  # dropout sits between to_out and the residual
            with ev("deslice"):
                out = ops.deslice_out_split(xs, sp, col_max, col_sum, S2, G, w_out, None if drop else residual)
            with ev("to_out"):
                out = self.to_out[1](out).to(x.dtype)
                if drop and residual is not None:
                    out = out + residual
            return out
        with ev("deslice"):
            out_x = (ops.deslice_split(xs, sp, col_max, col_sum, S2, G) if xs is not None
                     else ops.deslice(x_lin, sp, col_max, col_sum, S2p, H, C, G))  # [B, N, H*C]
            if padded:
                out_x, x_lin = ops.unpad_heads(out_x, H, Cr, C), x_real
        with ev("to_out"):
            out = self.to_out[1](self._linear("to_out", self.to_out[0], out_x)).to(x.dtype)
        if residual is not None:
            out = out + residual
        if return_intermediates:
            if xs is not None:
                x_lin = xs.to_float()
            return out, dict(x_lin=x_lin, col_max=col_max, col_sum=col_sum, eidetic=S, pos=pos,
                             processed=S2, out_x=out_x)
        return out
