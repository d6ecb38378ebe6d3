"""Torch-tensor front-ends of the C-ABI entry points (one call per stage).

These only validate, allocate outputs with torch (device memory + stream plumbing) and
hand raw pointers to libhaet_hip.so on the current HIP stream.  All tensors must live
on a ROCm device; there is no CPU path.
"""
from __future__ import annotations

import ctypes
import functools
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import HaetError, check

Tensor = torch.Tensor


def _stream() -> int:
    # raw hipStream_t of the current stream (torch.cuda.current_stream() builds a Stream object and
    # resolves the device through several Python layers: ~9 us per call, eight calls per forward)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _f32(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise HaetError(f"{name} must be a ROCm device tensor (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _p(t: Optional[Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def tree_slots(M: int) -> int:
    """Slots per cloud of the complete ball tree: 2 * 2^(ceil(log2 M) - 1) (balltree.pyx:58-60)."""
    depth = 0
    while (2 << depth) < M:
        depth += 1
    return 2 << depth


@functools.lru_cache(maxsize=None)
def _slice_c0(epsilon: float) -> float:
    # log(-log(eps)) evaluated in fp32 like the reference (IrregularMesh.py:172)
    return float(torch.log(-torch.log(torch.tensor(epsilon, dtype=torch.float32))))


class SliceParams:
    """Device views of the Rep-Slice / Ada-Temp parameters (in_project_slice, ada_temp_linear)."""

    def __init__(self, Ws: Tensor, bs: Tensor, wt: Tensor, bt: float, base_temp: float, epsilon: float):
        self.Ws, self.bs, self.wt = _f32(Ws, "in_project_slice.weight"), _f32(bs, "in_project_slice.bias"), _f32(wt.reshape(-1), "ada_temp_linear.weight")
        self.bt, self.base_temp = float(bt), float(base_temp)
        self.c0 = _slice_c0(float(epsilon))


def slice_partials(x_lin: Tensor, p: SliceParams, H: int, C: int, M: int) -> Tensor:
    """Slice pass -> [P, B*H*M*(C+2)] partial records (haet_slice_fwd)."""
    lib = _lib.load()
    x_lin = _f32(x_lin, "x_lin")
    B, N, HC = x_lin.shape
    if HC != H * C:
        raise HaetError(f"x_lin last dim {HC} != heads*dim_head {H * C}")
    P = lib.haet_slice_num_parts(B, N, H, C, M)
    if P < 0:
        check(P, "haet_slice_num_parts")
    rec = lib.haet_slice_part_floats(B, H, C, M)
    parts = torch.empty((P, rec), dtype=torch.float32, device=x_lin.device)
    check(lib.haet_slice_fwd(_p(x_lin), _p(p.Ws), _p(p.bs), _p(p.wt), p.bt, p.base_temp, p.c0,
                             B, N, H, C, M, _p(parts), _stream()), "haet_slice_fwd")
    return parts


def merge_records(records: Tensor, B: int, H: int, C: int, M: int, finalize: bool):
    """haet_slice_merge.  finalize=False -> one merged record [rec];
    finalize=True -> (col_max [B*H*M], col_sum [B*H*M], S [B*H*M, C])."""
    lib = _lib.load()
    records = _f32(records, "records")
    R = records.shape[0]
    dev = records.device
    if finalize:
        cm = torch.empty(B * H * M, dtype=torch.float32, device=dev)
        cs = torch.empty_like(cm)
        S = torch.empty((B * H * M, C), dtype=torch.float32, device=dev)
        check(lib.haet_slice_merge(_p(records), R, B, H, C, M, 1, _p(None), _p(cm), _p(cs), _p(S), _stream()),
              "haet_slice_merge")
        return cm, cs, S
    out = torch.empty(records.shape[1], dtype=torch.float32, device=dev)
    check(lib.haet_slice_merge(_p(records), R, B, H, C, M, 0, _p(out), _p(None), _p(None), _p(None), _stream()),
          "haet_slice_merge")
    return out


def token_pos(S: Tensor, d: int) -> Tensor:
    lib = _lib.load()
    S = _f32(S, "S")
    T, C = S.shape
    pos = torch.empty((T, d), dtype=torch.float32, device=S.device)
    check(lib.haet_token_pos(_p(S), T, C, d, _p(pos), _stream()), "haet_token_pos")
    return pos


def rot_targets(total_leaves: int, ball_sizes: Sequence[int]) -> List[int]:
    """balltree.pyx:643 (the TOTAL leaf count over all clouds enters -- reference quirk)."""
    return [max(0, int(math.log2(total_leaves / bs))) for bs in ball_sizes]


def balltree_build(pos: Optional[Tensor], n_clouds: int, M: int, strides: Sequence[int],
                   ball_sizes: Sequence[int], angle: float, tokens: Optional[Tensor] = None, d: int = 3):
    """Device build_balltree_with_rotations for equally sized clouds.
    Returns (tree_idx int32[n_clouds*Mp], tree_mask uint8[...], rot int32 flat | None, [level sizes]).
    With `tokens` [n_clouds*M, C] instead of `pos`, the positions are derived from the tokens' first d
    channels in the same launch (haet_tokens_tree_build) and returned as a fifth value."""
    lib = _lib.load()
    if tokens is not None:
        tokens = _f32(tokens, "tokens")
        pos = torch.empty(tokens.shape[0], d, dtype=torch.float32, device=tokens.device)
    else:
        pos = _f32(pos, "pos")
        d = pos.shape[1]
    if pos.shape[0] != n_clouds * M:
        raise HaetError("pos rows != n_clouds * M")
    if len(strides) != len(ball_sizes) - 1:
        raise AssertionError("Strides must be one less than ball sizes")
    Mp = tree_slots(M)
    nl = len(ball_sizes)
    dev = pos.device
    tree_idx = torch.empty(n_clouds * Mp, dtype=torch.int32, device=dev)
    tree_mask = torch.empty(n_clouds * Mp, dtype=torch.uint8, device=dev)
    sizes, n = [], Mp
    for i in range(nl):
        sizes.append(n_clouds * n)
        if i < nl - 1:
            n //= strides[i]
    rot = torch.empty(sum(sizes), dtype=torch.int32, device=dev) if angle > 0 else None
    tg = rot_targets(n_clouds * Mp, ball_sizes)
    c_str = (ctypes.c_int * max(1, nl - 1))(*list(strides)[: nl - 1]) if nl > 1 else (ctypes.c_int * 1)(1)
    c_tg = (ctypes.c_int * nl)(*tg)
    if tokens is not None:
        check(lib.haet_tokens_tree_build(_p(tokens), tokens.shape[1], _p(pos), n_clouds, M, d, nl, c_str, c_tg,
                                         float(angle), _p(tree_idx), _p(tree_mask), _p(rot), _stream()),
              "haet_tokens_tree_build")
        return tree_idx, tree_mask, rot, sizes, pos
    check(lib.haet_balltree_build(_p(pos), n_clouds, M, d, nl, c_str, c_tg, float(angle),
                                  _p(tree_idx), _p(tree_mask), _p(rot), _stream()), "haet_balltree_build")
    return tree_idx, tree_mask, rot, sizes


def check_barriers(states):
    """ErwinTrainState.check_barrier for several states with ONE device -> host copy."""
    states = [st for st in states if st.workspace.is_cuda]
    if not states:
        return
    words = torch.cat([st.workspace[:2] for st in states]).cpu().contiguous()
    lib = _lib.load()
    for i, st in enumerate(states):
        rc = lib.haet_erwin_barrier_status(ctypes.c_void_p(words.data_ptr() + 8 * i))
        if rc != 0:
            st.workspace[1:2].zero_()  # re-arm
            check(rc, "haet_erwin_barrier_status")


class ErwinTrainState:
    """Device plumbing of train-mode BatchNorm for haet_erwin_fwd / haet_token_stage_fwd (include/haet_hip.h
    haet_erwin_train): running statistics of the pool / unpool norms (updated in place by the kernel), the
    batch statistics of the forward (kept for the backward) and the grid-barrier workspace.  `set_tape` attaches
    the tape the backward kernel reads (also used, without the BatchNorm part, in eval mode)."""

    def set_tape(self, tape: Optional[Tensor], op_floats: int = 0):
        self.tape = tape
        self.c.tape = 0 if tape is None else tape.data_ptr()
        self.c.tape_op_floats = int(op_floats)

    def check_barrier(self):
        """Read the workspace's barrier words (a device -> host copy: call at a point that synchronises anyway) and
        raise HaetError if a grid barrier of a train-mode launch timed out since the last check (include/haet_hip.h,
        residency contract: the launch's workgroups must all be resident, i.e. the device otherwise idle)."""
        words = self.workspace[:2].cpu().contiguous()
        rc = _lib.load().haet_erwin_barrier_status(ctypes.c_void_p(words.data_ptr()))
        if rc != 0:
            self.workspace[1:2].zero_()  # re-arm
            check(rc, "haet_erwin_barrier_status")

    def __init__(self, erwin, desc, n_clouds: int, device):
        lib = _lib.load()
        self.c = _lib.ErwinTrain()
        self.keep = []
        self.tape = None
        n_levels = len(erwin.encoder) + 1
        cmax = max(erwin.c_hidden)
        self.batch_stats = torch.zeros((2 * _lib.MAX_LEVELS, 2, cmax), dtype=torch.float32, device=device)
        self.workspace = torch.zeros(lib.haet_erwin_train_workspace_bytes(ctypes.byref(desc), n_clouds) // 4 + 1,
                                     dtype=torch.float32, device=device)
        self.c.batch_stats = self.batch_stats.data_ptr()
        self.c.workspace = self.workspace.data_ptr()
        for i, layer in enumerate(erwin.encoder):
            self._slot(2 * i, layer.pool.norm)
        for k, layer in enumerate(erwin.decoder):
            self._slot(2 * (n_levels - 2 - k) + 1, layer.unpool.norm)

    def _slot(self, slot, bn):
        if bn.momentum is None:
            raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative average) is not supported by the HIP Erwin stage")
        if bn.track_running_stats and bn.running_mean is not None:
            for t in (bn.running_mean, bn.running_var):
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise HaetError("BatchNorm running statistics must be contiguous fp32")
            self.c.running_mean[slot] = bn.running_mean.data_ptr()
            self.c.running_var[slot] = bn.running_var.data_ptr()
            self.c.num_batches_tracked[slot] = bn.num_batches_tracked.data_ptr()
            self.keep += [bn.running_mean, bn.running_var, bn.num_batches_tracked]


def erwin_fwd(tokens: Tensor, pos: Tensor, tree_idx: Tensor, tree_mask: Tensor, rot: Optional[Tensor],
              weights: Tensor, desc: "_lib.ErwinDesc", n_clouds: int, M: int, train: Optional[ErwinTrainState] = None) -> Tensor:
    lib = _lib.load()
    tokens, pos = _f32(tokens, "tokens"), _f32(pos, "pos")
    out = torch.empty_like(tokens)
    check(lib.haet_erwin_fwd(_p(tokens), _p(pos), _p(tree_idx), _p(tree_mask), _p(rot), _p(weights),
                             weights.numel(), ctypes.byref(desc), ctypes.byref(train.c) if train is not None else None,
                             n_clouds, M, _p(out), _stream()),
          "haet_erwin_fwd")
    return out


def erwin_tape(desc: "_lib.ErwinDesc", n_clouds: int, M: int, device):
    """(tape tensor [n_clouds, n_ops * op_floats], op_floats) for the forward to record and haet_erwin_bwd to read."""
    lib = _lib.load()
    n_ops, opf = ctypes.c_int(0), ctypes.c_int(0)
    per = lib.haet_erwin_tape_floats(ctypes.byref(desc), M, ctypes.byref(n_ops), ctypes.byref(opf))
    return torch.empty((n_clouds, per), dtype=torch.float32, device=device), opf.value


def erwin_bwd_supported(desc: "_lib.ErwinDesc", n_clouds: int, M: int) -> bool:
    return bool(_lib.load().haet_erwin_bwd_supported(ctypes.byref(desc), n_clouds, M))


def erwin_bwd(g_out: Tensor, tape: Tensor, pos: Tensor, tree_idx: Tensor, tree_mask: Tensor, rot: Optional[Tensor],
              blob_f32: Tensor, total: int, blob_fwd: Tensor, desc: "_lib.ErwinDesc", n_clouds: int, M: int,
              train: Optional[ErwinTrainState] = None):
    """haet_erwin_bwd -> (dL/d tokens [n_clouds*M, c0], grads [n_clouds, total]: every parameter gradient per cloud,
    at the offsets of the fp32 blob's layout)."""
    lib = _lib.load()
    g_out = _f32(g_out, "g_out")
    g_in = torch.empty_like(g_out)
    grads = torch.zeros((n_clouds, total), dtype=torch.float32, device=g_out.device)
    check(lib.haet_erwin_bwd(_p(g_out), _p(tape), _p(pos), _p(tree_idx), _p(tree_mask), _p(rot), _p(blob_f32), _p(blob_fwd),
                             ctypes.byref(desc), _p(train.batch_stats if train is not None else None),
                             _p(train.workspace if train is not None else None), n_clouds, M, _p(g_in), _p(grads),
                             _stream()), "haet_erwin_bwd")
    return g_in, grads


def token_stage_supported(desc: "_lib.ErwinDesc", n_clouds: int, M: int) -> bool:
    return bool(_lib.load().haet_token_stage_supported(ctypes.byref(desc), n_clouds, M))


def erwin_frags(weights: Tensor, desc: "_lib.ErwinDesc", M: int) -> Optional[Tensor]:
    """The Linear weights of an Erwin stage in MFMA fragment order for the fragment kernel (haet_erwin_frag_pack), or
    None when (desc, M) is outside that kernel's envelope.  Built on the device from the packed blob; the caller
    caches it per weight version."""
    lib = _lib.load()
    nbytes = lib.haet_erwin_frag_bytes(ctypes.byref(desc), M)
    if not nbytes:
        return None
    frags = torch.empty(nbytes // 4, dtype=torch.float32, device=weights.device)
    check(lib.haet_erwin_frag_pack(_p(weights), weights.numel(), ctypes.byref(desc), M, _p(frags), nbytes, _stream()),
          "haet_erwin_frag_pack")
    return frags


def token_stage(S: Tensor, weights: Tensor, desc: "_lib.ErwinDesc", n_clouds: int, M: int, d: int, strides: Sequence[int],
                angle: float, train: Optional[ErwinTrainState] = None, want_tree: bool = False,
                frags: Optional[Tensor] = None):
    """haet_token_stage_fwd: positions + ball tree + Erwin in one launch.  Returns (tokens', pos) and, with
    want_tree, also (tree_idx, tree_mask, rot flat | None, level sizes).  With `frags` (erwin_frags; eval mode, no
    tape) the launch is the fragment kernel, haet_token_stage_fwd_frag."""
    lib = _lib.load()
    S = _f32(S, "S")
    dev = S.device
    out = torch.empty_like(S)
    pos = torch.empty((S.shape[0], d), dtype=torch.float32, device=dev)
    tree_idx = tree_mask = rot = None
    sizes = []
    if want_tree:
        tree_idx = torch.empty(n_clouds * M, dtype=torch.int32, device=dev)
        tree_mask = torch.empty(n_clouds * M, dtype=torch.uint8, device=dev)
        n = M
        for i in range(desc.n_levels):
            sizes.append(n_clouds * n)
            if i < desc.n_levels - 1:
                n //= strides[i]
        rot = torch.empty(sum(sizes), dtype=torch.int32, device=dev) if angle > 0 else None
    if frags is not None and train is None:
        check(lib.haet_token_stage_fwd_frag(_p(S), n_clouds, M, float(angle), _p(weights), weights.numel(), _p(frags),
                                            frags.numel() * 4, ctypes.byref(desc), _p(pos), _p(tree_idx), _p(tree_mask),
                                            _p(rot), _p(out), _stream()), "haet_token_stage_fwd_frag")
    else:
        check(lib.haet_token_stage_fwd(_p(S), n_clouds, M, float(angle), _p(weights), weights.numel(), ctypes.byref(desc),
                                       ctypes.byref(train.c) if train is not None else None, _p(pos), _p(tree_idx),
                                       _p(tree_mask), _p(rot), _p(out), _stream()), "haet_token_stage_fwd")
    if want_tree:
        return out, pos, tree_idx, tree_mask, rot, sizes
    return out, pos


def deslice(x_lin: Tensor, p: SliceParams, col_max: Tensor, col_sum: Tensor, S2: Tensor,
            H: int, C: int, M: int) -> Tensor:
    lib = _lib.load()
    x_lin = _f32(x_lin, "x_lin")
    B, N, _ = x_lin.shape
    out = torch.empty_like(x_lin)
    check(lib.haet_deslice_fwd(_p(x_lin), _p(p.Ws), _p(p.bs), _p(p.wt), p.bt, p.base_temp, p.c0,
                               _p(col_max), _p(col_sum), _p(_f32(S2, "S2")), B, N, H, C, M, _p(out),
                               _stream()), "haet_deslice_fwd")
    return out


# --------------------------------------------------------------------------- backward of the two passes
def deslice_bwd_tokens(x_lin: Tensor, p: SliceParams, col_max: Tensor, col_sum: Tensor, g_out: Tensor,
                       H: int, C: int, M: int) -> Tensor:
    """dS2 [B*H*M, C] = sum_n w[n,g] g_out[n]  (haet_deslice_bwd_tokens + the sum of its partials)."""
    lib = _lib.load()
    x_lin, g_out = _f32(x_lin, "x_lin"), _f32(g_out, "g_out")
    B, N, _ = x_lin.shape
    P = lib.haet_slice_bwd_num_parts(B, N, H, C, M)
    if P < 0:
        check(P, "haet_slice_bwd_num_parts")
    parts = torch.empty((P, B * H * M, C), dtype=torch.float32, device=x_lin.device)
    check(lib.haet_deslice_bwd_tokens(_p(x_lin), _p(p.Ws), _p(p.bs), _p(p.wt), p.bt, p.base_temp, p.c0,
                                      _p(col_max), _p(col_sum), _p(g_out), B, N, H, C, M, _p(parts), _stream()),
          "haet_deslice_bwd_tokens")
    return parts.sum(dim=0)


def slice_bwd_points(x_lin: Tensor, p: SliceParams, col_max: Tensor, col_sum: Tensor, g_out: Tensor,
                     S2: Tensor, dS: Tensor, A: Tensor, H: int, C: int, M: int):
    """haet_slice_bwd_points -> (dx_lin [B,N,H*C], dWs [M,C], dbs [M], dwt [C], dbt [])."""
    lib = _lib.load()
    x_lin, g_out = _f32(x_lin, "x_lin"), _f32(g_out, "g_out")
    S2, dS, A = _f32(S2, "S2"), _f32(dS, "dS"), _f32(A, "A")
    B, N, _ = x_lin.shape
    P = lib.haet_slice_bwd_num_parts(B, N, H, C, M)
    if P < 0:
        check(P, "haet_slice_bwd_num_parts")
    R = lib.haet_slice_bwd_param_floats(C, M)
    dx = torch.empty_like(x_lin)
    pp = torch.empty((P * B * H, R), dtype=torch.float32, device=x_lin.device)
    check(lib.haet_slice_bwd_points(_p(x_lin), _p(p.Ws), _p(p.bs), _p(p.wt), p.bt, p.base_temp, p.c0,
                                    _p(col_max), _p(col_sum), _p(g_out), _p(S2), _p(dS), _p(A), B, N, H, C, M,
                                    _p(dx), _p(pp), _stream()), "haet_slice_bwd_points")
    g = pp.sum(dim=0)
    return dx, g[:M * C].view(M, C), g[M * C:M * C + M], g[M * C + M:M * C + M + C], g[M * C + M + C]


class SplitLinear:
    """A Linear layer's weight as bf16 hi/lo planes for haet_linear_fwd (built once per weight
    version by haet_split_bf16 on the device)."""

    def __init__(self, weight: Tensor, bias: Optional[Tensor]):
        lib = _lib.load()
        w = _f32(weight.detach(), "weight")
        self.O, self.K = w.shape
        self.tiled = self.O % 16 == 0 and self.K % 8 == 0
        if self.tiled:  # plane + its tile-ordered copy (what the Linear entry points expect: include/haet_hip.h, ABI 9)
            self.hi_all = torch.empty(2 * self.O * self.K, dtype=torch.int16, device=w.device)
            self.lo_all = torch.empty_like(self.hi_all)
            check(lib.haet_split_bf16_tiled(_p(w), self.O, self.K, _p(self.hi_all), _p(self.lo_all), _stream()),
                  "haet_split_bf16_tiled")
            self.hi = self.hi_all[:self.O * self.K].view(self.O, self.K)  # (same base pointers: `_p(w.hi)` is the plane,
            self.lo = self.lo_all[:self.O * self.K].view(self.O, self.K)  #  the copy follows it in memory)
        else:  # (shapes no Linear kernel takes: planes only)
            self.hi = torch.empty((self.O, self.K), dtype=torch.int16, device=w.device)
            self.lo = torch.empty_like(self.hi)
            check(lib.haet_split_bf16(_p(w), w.numel(), _p(self.hi), _p(self.lo), _stream()), "haet_split_bf16")
        self.bias = None if bias is None else _f32(bias.detach(), "bias")


def linear_supported(K: int, O: int) -> bool:
    return bool(_lib.load().haet_linear_supported(int(K), int(O)))


def linear_ln_supported(K: int, O: int, rows: int) -> bool:
    """LayerNorm fused into the GEMM's loader: the W-stationary kernel's envelope."""
    return project_split_supported(K, O, rows)


def linear(x: Tensor, w: SplitLinear, gelu: bool = False, ln=None) -> Tensor:
    """out[..., O] = x[..., K] @ W^T + b on the matrix cores (split-bf16, fp32-class accuracy), optionally
    followed by GELU (erf) in the epilogue; `ln` = (weight, bias, eps) applies nn.LayerNorm(K) to x in the loader."""
    lib = _lib.load()
    x = _f32(x, "x")
    rows = x.numel() // w.K
    out = torch.empty(x.shape[:-1] + (w.O,), dtype=torch.float32, device=x.device)
    if ln is not None:
        check(lib.haet_linear_ln_fwd(_p(x), _p(_f32(ln[0].detach(), "ln.weight")), _p(_f32(ln[1].detach(), "ln.bias")),
                                     float(ln[2]), _p(w.hi), _p(w.lo), _p(w.bias), rows, w.K, w.O, 1 if gelu else 0,
                                     _p(out), _stream()), "haet_linear_ln_fwd")
    elif gelu:
        check(lib.haet_linear_act_fwd(_p(x), _p(w.hi), _p(w.lo), _p(w.bias), rows, w.K, w.O, 1, _p(out), _stream()),
              "haet_linear_act_fwd")
    else:
        check(lib.haet_linear_fwd(_p(x), _p(w.hi), _p(w.lo), _p(w.bias), rows, w.K, w.O, _p(out), _stream()),
              "haet_linear_fwd")
    return out


# --------------------------------------------------------------------------- the block around the path
def layernorm_supported(D: int) -> bool:
    return bool(_lib.load().haet_layernorm_supported(int(D)))


def layernorm(x: Tensor, weight: Tensor, bias: Tensor, eps: float) -> Tensor:
    """nn.LayerNorm over the last dimension (haet_layernorm_fwd)."""
    lib = _lib.load()
    x = _f32(x, "x")
    D = x.shape[-1]
    out = torch.empty_like(x)
    check(lib.haet_layernorm_fwd(_p(x), _p(_f32(weight.detach(), "weight")), _p(_f32(bias.detach(), "bias")), float(eps),
                                 x.numel() // D, D, _p(out), _stream()), "haet_layernorm_fwd")
    return out


def add_layernorm(x: Tensor, res: Tensor, weight: Tensor, bias: Tensor, eps: float):
    """(x + res, LayerNorm(x + res)) in one pass (haet_add_layernorm_fwd)."""
    lib = _lib.load()
    x, res = _f32(x, "x"), _f32(res, "res")
    D = x.shape[-1]
    s, out = torch.empty_like(x), torch.empty_like(x)
    check(lib.haet_add_layernorm_fwd(_p(x), _p(res), _p(_f32(weight.detach(), "weight")), _p(_f32(bias.detach(), "bias")),
                                     float(eps), x.numel() // D, D, _p(s), _p(out), _stream()), "haet_add_layernorm_fwd")
    return s, out


def ln_linear_small_supported(D: int, OD: int) -> bool:
    return bool(_lib.load().haet_ln_linear_small_supported(int(D), int(OD)))


def ln_linear_small(x: Tensor, res: Optional[Tensor], ln_weight: Tensor, ln_bias: Tensor, eps: float,
                    weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    """Linear(LayerNorm(x (+ res))) with at most 16 outputs, one pass (haet_ln_linear_small_fwd)."""
    lib = _lib.load()
    x = _f32(x, "x")
    D = x.shape[-1]
    OD = weight.shape[0]
    out = torch.empty(x.shape[:-1] + (OD,), dtype=torch.float32, device=x.device)
    check(lib.haet_ln_linear_small_fwd(_p(x), _p(None if res is None else _f32(res, "res")),
                                       _p(_f32(ln_weight.detach(), "ln.weight")), _p(_f32(ln_bias.detach(), "ln.bias")),
                                       float(eps), _p(_f32(weight.detach(), "weight")),
                                       _p(None if bias is None else _f32(bias.detach(), "bias")), x.numel() // D, D, OD,
                                       _p(out), _stream()), "haet_ln_linear_small_fwd")
    return out


def layernorm_bwd(x: Tensor, weight: Tensor, gy: Tensor, eps: float):
    """(dx, dgamma, dbeta) of nn.LayerNorm over the last dimension (haet_layernorm_bwd + the sum of its partials)."""
    lib = _lib.load()
    x, gy = _f32(x, "x"), _f32(gy, "gy")
    D = x.shape[-1]
    rows = x.numel() // D
    P = lib.haet_layernorm_bwd_num_parts(rows)
    dx = torch.empty_like(x)
    parts = torch.empty((P, 2 * D), dtype=torch.float32, device=x.device)
    check(lib.haet_layernorm_bwd(_p(x), _p(_f32(weight.detach(), "weight")), _p(gy), float(eps), rows, D, _p(dx),
                                 _p(parts), _stream()), "haet_layernorm_bwd")
    s = parts.sum(dim=0)
    return dx, s[:D], s[D:]


def linear_narrow_supported(K: int, O: int) -> bool:
    return bool(_lib.load().haet_linear_narrow_supported(int(K), int(O)))


def linear_narrow(x: Tensor, weight: Tensor, bias: Optional[Tensor], gelu: bool = False) -> Tensor:
    """Linear (+ GELU) with at most 16 input features (haet_linear_narrow_fwd)."""
    lib = _lib.load()
    x = _f32(x, "x")
    O, K = weight.shape
    out = torch.empty(x.shape[:-1] + (O,), dtype=torch.float32, device=x.device)
    check(lib.haet_linear_narrow_fwd(_p(x), _p(_f32(weight.detach(), "weight")),
                                     _p(None if bias is None else _f32(bias.detach(), "bias")), x.numel() // K, K, O,
                                     1 if gelu else 0, _p(out), _stream()), "haet_linear_narrow_fwd")
    return out


def linear_wgrad(g: Tensor, x: Tensor):
    """(dW [O,K], db [O]) of a Linear from g = dL/d output [..., O] and its input x [..., K], contracted over
    all leading dimensions: haet_linear_wgrad where O, K are multiples of 128, the library GEMM otherwise."""
    lib = _lib.load()
    g, x = _f32(g, "g"), _f32(x, "x")
    O, K = g.shape[-1], x.shape[-1]
    g2, x2 = g.reshape(-1, O), x.reshape(-1, K)
    N = g2.shape[0]
    if x2.shape[0] != N:
        raise HaetError("linear_wgrad: g and x disagree on the number of rows")
    if not lib.haet_linear_wgrad_supported(O, K):
        return g2.t() @ x2, g2.sum(dim=0)
    P = lib.haet_linear_wgrad_num_parts(N)
    ob, kb = (O + 255) // 256, (K + 255) // 256
    parts = torch.empty((P, ob * kb, 256 * 256 + 256), dtype=torch.float32, device=g.device)
    check(lib.haet_linear_wgrad(_p(g2), _p(x2), N, O, K, _p(parts), _stream()), "haet_linear_wgrad")
    # (a record's tile is written only where it overlaps the O x K matrix: sum that part, not the whole 256 x 256)
    db = parts[:, :, 256 * 256:].sum(dim=0).view(ob, kb, 256)[:, 0].reshape(ob * 256)[:O]
    if ob * kb == 1:
        return parts[:, 0, :256 * 256].view(P, 256, 256)[:, :O, :K].sum(dim=0), db
    s = parts[:, :, :256 * 256].sum(dim=0)
    dW = s.view(ob, kb, 256, 256).permute(0, 2, 1, 3).reshape(ob * 256, kb * 256)[:O, :K]
    return dW, db


def conv_wgrad_supported(Cin: int, O: int) -> bool:
    return bool(_lib.load().haet_linear_wgrad_supported(int(O), int(Cin)))


def conv_wgrad(g: Tensor, x: Tensor, grid: Tuple[int, ...], kernel: Tuple[int, ...]):
    """(dW [O, C_in, *kernel], db [O]) of a same-padded stride-1 Conv2d / Conv3d over the channels-last grid:
    x [B, N, C_in], g = dL/d output [B, N, O]; one haet_conv_wgrad_tap launch per kernel tap."""
    lib = _lib.load()
    g, x = _f32(g, "g"), _f32(x, "x")
    B, N, O = g.shape
    Cin = x.shape[-1]
    gH, gW = grid[0], grid[1]
    gD = grid[2] if len(grid) > 2 else 1
    ks = tuple(kernel) + (1,) * (3 - len(kernel))
    if gH * gW * gD != N or x.shape[1] != N:
        raise HaetError("conv_wgrad: tensors do not match the grid")
    P = lib.haet_linear_wgrad_num_parts(B * N)
    ob, kb = (O + 255) // 256, (Cin + 255) // 256
    taps = [(a, b, c) for a in range(ks[0]) for b in range(ks[1]) for c in range(ks[2])]
    rec = 256 * 256 + 256
    # the taps' split-K partials side by side, ONE reduction and one scatter into dW per group of taps (a launch-bound
    # regime at the PDE benchmarks' grids: 3 launches per tap were 27 of a layer's backward); groups of <= 256 MB
    group = max(1, min(len(taps), (256 << 20) // (P * ob * kb * rec * 4)))
    parts = torch.empty((group, P, ob * kb, rec), dtype=torch.float32, device=g.device)
    dW = torch.empty((O, Cin, len(taps)), dtype=torch.float32, device=g.device)
    db = None
    ro, rc = min(O, 256), min(Cin, 256)  # (one tile: only its O x Cin corner is summed)
    for t0 in range(0, len(taps), group):
        n = min(group, len(taps) - t0)
        for i, (a, b, c) in enumerate(taps[t0:t0 + n]):
            check(lib.haet_conv_wgrad_tap(_p(g), _p(x), B, gH, gW, gD, O, Cin, a - ks[0] // 2, b - ks[1] // 2,
                                          c - ks[2] // 2, _p(parts[i]), _stream()), "haet_conv_wgrad_tap")
        tiles = parts[:n, :, :, :256 * 256].view(n, P, ob, kb, 256, 256)
        if ob * kb == 1:
            s = tiles[:, :, 0, 0, :ro, :rc].sum(dim=1)                                  # [n, O, Cin]
        else:
            s = tiles.sum(dim=1).permute(0, 1, 3, 2, 4).reshape(n, ob * 256, kb * 256)[:, :O, :Cin]
        dW[:, :, t0:t0 + n] = s.permute(1, 2, 0)
        if db is None:
            db = parts[0, :, :, 256 * 256:].sum(dim=0).view(ob, kb, 256)[:, 0].reshape(ob * 256)[:O].clone()
    return dW.reshape((O, Cin) + tuple(kernel)), db


# --------------------------------------------------------------------------- fused operand format
class SplitPoints:
    """x_lin in the streaming passes' operand format (haet_project_split_fwd): `cells` [B, N, H*C]
    4-byte cells holding, per 4 channels, 4 bf16 hi then 4 bf16 lo, and `s` [B, N, H] = log2(e)/tau."""

    def __init__(self, cells: Tensor, s: Tensor, H: int, C: int):
        self.cells, self.s, self.H, self.C = cells, s, H, C

    def to_float(self) -> Tensor:
        """fp32 x_lin = hi + lo (for inspection / tests; 2^-17 relative to the fp32 projection)."""
        B, N, HC = self.cells.shape
        w = self.cells.view(torch.bfloat16).reshape(B, N, HC // 4, 2, 4).float()
        return (w[:, :, :, 0] + w[:, :, :, 1]).reshape(B, N, HC)


def project_split_supported(K: int, O: int, rows: int) -> bool:
    return bool(_lib.load().haet_project_split_supported(int(K), int(O), int(rows)))


def slice_mfma_supported(B: int, N: int, H: int, C: int, M: int) -> bool:
    """Whether the fp32-input passes run this shape on the split-bf16 matrix-core kernels (haet_slice_mfma_supported)."""
    return bool(_lib.load().haet_slice_mfma_supported(B, N, H, C, M))


def pad_heads(x: Tensor, H: int, C: int, Cp: int) -> Tensor:
    """[..., H*C] -> [..., H*Cp]: every head's C channels followed by Cp - C zeros."""
    lead = x.shape[:-1]
    return torch.nn.functional.pad(x.reshape(lead + (H, C)), (0, Cp - C)).reshape(lead + (H * Cp,))


def unpad_heads(x: Tensor, H: int, C: int, Cp: int) -> Tensor:
    """Inverse of pad_heads (a copy)."""
    lead = x.shape[:-1]
    return x.reshape(lead + (H, Cp))[..., :C].reshape(lead + (H * C,))


def slice_split_supported(B: int, N: int, H: int, C: int, M: int) -> bool:
    return bool(_lib.load().haet_slice_split_supported(B, N, H, C, M))


def project_split(x: Tensor, w: SplitLinear, p: SliceParams, H: int, C: int, ln=None) -> SplitPoints:
    """in_project_x on the matrix cores, output in the split operand format + s = log2(e)/tau;
    `ln` = (weight, bias, eps): nn.LayerNorm over x's last dimension fused into the loader."""
    lib = _lib.load()
    x = _f32(x, "x")
    B, N, K = x.shape
    cells = torch.empty((B, N, w.O), dtype=torch.float32, device=x.device)
    s = torch.empty((B, N, H), dtype=torch.float32, device=x.device)
    if ln is not None:
        check(lib.haet_project_split_ln_fwd(_p(x), _p(_f32(ln[0].detach(), "ln.weight")), _p(_f32(ln[1].detach(), "ln.bias")),
                                            float(ln[2]), _p(w.hi), _p(w.lo), _p(w.bias), _p(p.wt), p.bt, p.base_temp,
                                            B * N, K, w.O, _p(cells), _p(s), _stream()), "haet_project_split_ln_fwd")
        return SplitPoints(cells, s, H, C)
    check(lib.haet_project_split_fwd(_p(x), _p(w.hi), _p(w.lo), _p(w.bias), _p(p.wt), p.bt, p.base_temp,
                                     B * N, K, w.O, _p(cells), _p(s), _stream()), "haet_project_split_fwd")
    return SplitPoints(cells, s, H, C)


def slice_partials_split(xs: SplitPoints, p: SliceParams, M: int) -> Tensor:
    lib = _lib.load()
    B, N, _ = xs.cells.shape
    H, C = xs.H, xs.C
    P = lib.haet_slice_num_parts(B, N, H, C, M)
    if P < 0:
        check(P, "haet_slice_num_parts")
    parts = torch.empty((P, lib.haet_slice_part_floats(B, H, C, M)), dtype=torch.float32, device=xs.cells.device)
    check(lib.haet_slice_fwd_split(_p(xs.cells), _p(xs.s), _p(p.Ws), _p(p.bs), p.c0, B, N, H, C, M, _p(parts),
                                   _stream()), "haet_slice_fwd_split")
    return parts


def deslice_split(xs: SplitPoints, p: SliceParams, col_max: Tensor, col_sum: Tensor, S2: Tensor, M: int) -> Tensor:
    lib = _lib.load()
    B, N, _ = xs.cells.shape
    out = torch.empty_like(xs.cells)
    check(lib.haet_deslice_fwd_split(_p(xs.cells), _p(xs.s), _p(p.Ws), _p(p.bs), p.c0, _p(col_max), _p(col_sum),
                                     _p(_f32(S2, "S2")), B, N, xs.H, xs.C, M, _p(out), _stream()),
          "haet_deslice_fwd_split")
    return out


def deslice_out_supported(H: int, C: int, M: int, O: int) -> bool:
    return bool(_lib.load().haet_deslice_out_supported(int(H), int(C), int(M), int(O)))


def deslice_out_split(xs: SplitPoints, p: SliceParams, col_max: Tensor, col_sum: Tensor, S2: Tensor, M: int,
                      w: SplitLinear, residual: Optional[Tensor] = None) -> Tensor:
    """to_out(de-slice(...)) in one pass (haet_deslice_out_fwd_split): the [B, N, H*C] tensor between them is never
    written.  `w`: the to_out Linear as split planes [O][H*C]; `residual` [B, N, O] is added to the output."""
    lib = _lib.load()
    B, N, _ = xs.cells.shape
    out = torch.empty((B, N, w.O), dtype=torch.float32, device=xs.cells.device)
    check(lib.haet_deslice_out_fwd_split(_p(xs.cells), _p(xs.s), _p(p.Ws), _p(p.bs), p.c0, _p(col_max), _p(col_sum),
                                         _p(_f32(S2, "S2")), _p(w.hi), _p(w.lo), _p(w.bias),
                                         _p(None if residual is None else _f32(residual, "residual")), B, N, xs.H, xs.C, M, w.O,
                                         _p(out), _stream()), "haet_deslice_out_fwd_split")
    return out


def conv_project_split(x: Tensor, w: "SplitConv", grid: Tuple[int, ...], p: SliceParams, H: int, C: int, ln=None) -> SplitPoints:
    """The convolution projection straight into the streaming passes' operand format + s = log2(e)/tau
    (haet_conv_project_split_fwd; dim_head C = 32)."""
    lib = _lib.load()
    x = _f32(x, "x")
    B, N, Cin = x.shape
    gH, gW = grid[0], grid[1]
    gD = grid[2] if len(grid) > 2 else 1
    if gH * gW * gD != N or Cin != w.Cin or w.O != H * C or C != 32:
        raise HaetError("conv_project_split: x does not match the grid / weight (dim_head must be 32)")
    if ln is not None or conv_cells_applies(Cin, w.O, B * N, w.taps):
        return conv_project_cells_split(x, w, grid, p, H, C, ln)
    cells = torch.empty((B, N, w.O), dtype=torch.float32, device=x.device)
    s = torch.empty((B, N, H), dtype=torch.float32, device=x.device)
    kH, kW, kD = w.kernel
    if _lib.precision() == "fp16x2" and w.f16_ok():
        check(lib.haet_conv_project_split_f16_fwd(_p(x), _p(w.f16()), _p(w.lin.bias), _p(p.wt), p.bt, p.base_temp, B, gH, gW, gD,
                                                  Cin, w.O, kH, kW, kD, _p(cells), _p(s), _stream()),
              "haet_conv_project_split_f16_fwd")
        return SplitPoints(cells, s, H, C)
    check(lib.haet_conv_project_split_fwd(_p(x), _p(w.lin.hi), _p(w.lin.lo), _p(w.lin.bias), _p(p.wt), p.bt, p.base_temp,
                                          B, gH, gW, gD, Cin, w.O, kH, kW, kD, _p(cells), _p(s), _stream()),
          "haet_conv_project_split_fwd")
    return SplitPoints(cells, s, H, C)


# --------------------------------------------------------------------------- structured projection
class SplitConv:
    """A Conv2d / Conv3d weight [O, C_in, k, k(, k)] as split planes laid out [O][C_in/32][taps][32]
    (the implicit GEMM's W, haet_conv_project_fwd)."""

    def __init__(self, weight: Tensor, bias: Optional[Tensor]):
        w = weight.detach()
        self.kernel = tuple(w.shape[2:]) + (1,) * (5 - w.dim())      # (kH, kW, kD)
        self.Cin, self.O = w.shape[1], w.shape[0]
        taps = 1
        for k in w.shape[2:]:
            taps *= k
        # [O, C_in, taps] -> [O, C_in/32, 32, taps] -> [O, C_in/32, taps, 32]: channel block, tap, channel
        flat = (w.reshape(self.O, self.Cin // 32, 32, taps).permute(0, 1, 3, 2).reshape(self.O, -1).contiguous())
        self.lin = SplitLinear(flat, bias)
        self.taps = taps
        self._f16 = None

    def _f16_make(self) -> None:
        w = self.lin.hi.view(torch.bfloat16).float() + self.lin.lo.view(torch.bfloat16).float()
        # fp16 range guard (one reduction per weight version): beyond 65504 the single plane would hold inf, and the
        # activations' fp16 hi + lo split inherits the same range -- such a layer stays on bf16x3
        self._f16_in_range = bool(torch.isfinite(w).all()) and float(w.abs().max()) <= 65504.0
        self._f16 = w.half().contiguous() if self._f16_in_range else False

    def f16_ok(self) -> bool:
        """Whether the opt-in fp16x2 arithmetic may serve this weight (all |w| inside the fp16 range)."""
        if self._f16 is None:
            self._f16_make()
        return self._f16 is not False

    def f16(self) -> Tensor:
        """The weight as ONE fp16 plane (same layout), for the opt-in fp16x2 arithmetic; made on first use."""
        if not self.f16_ok():
            raise HaetError("fp16x2: weight outside the fp16 range")
        return self._f16


# --------------------------------------------------------------------------- LDS-DMA convolution (csrc/conv8.hip)
CONV_CELLS_MIN_ROWS = 49152  # 192 row tiles of 256: below that the 128-row tiles of the streaming kernel fill the chip better (tools/conv8_sweep.py: 0.86-0.97x at 128 tiles, 1.3-1.6x from 256)


def conv_cells_applies(Cin: int, O: int, rows: int, taps: int = 27) -> bool:
    """Whether the LDS-DMA implicit GEMM (haet_conv_project_cells_fwd: x pre-split into k-tile cells once per forward)
    serves this convolution: O = 256, C_in % 32 = 0, at most 32 kernel taps, a grid large enough to fill the chip (both
    arithmetic modes: bf16x3 and the opt-in fp16x2)."""
    if os.environ.get("HAET_CONV_V1") or rows < CONV_CELLS_MIN_ROWS or taps > 32:
        return False
    return bool(_lib.load().haet_conv_cells_supported(int(Cin), int(O)))


def _conv_f16(w: "SplitConv") -> bool:
    return _lib.precision() == "fp16x2" and w.f16_ok()


def conv_presplit(x: Tensor, ln=None, f16: bool = False) -> Tensor:
    """x [B, N, C_in] fp32 -> the convolution's operand cells [(B*N + 1), C_in] (4-byte slots: per row and 32-channel block
    {32 x bf16 hi | 32 x bf16 lo}; the extra last row is zeros = the padding every out-of-grid tap reads).
    `ln` = (weight, bias, eps): nn.LayerNorm(C_in) applied to every row first (C_in = 256)."""
    lib = _lib.load()
    x = _f32(x, "x")
    B, N, Cin = x.shape
    cells = torch.empty((B * N + 1, Cin), dtype=torch.float32, device=x.device)
    g = b = None
    eps = 0.0
    if ln is not None:
        g, b, eps = _f32(ln[0].detach(), "ln.weight"), _f32(ln[1].detach(), "ln.bias"), float(ln[2])
    check(lib.haet_conv_presplit(_p(x), B * N, Cin, _p(cells), _p(g), _p(b), eps, int(f16), _stream()), "haet_conv_presplit")
    return cells


def _grid3(grid, N):
    gH, gW = grid[0], grid[1]
    gD = grid[2] if len(grid) > 2 else 1
    if gH * gW * gD != N:
        raise HaetError("convolution: x does not match the grid")
    return gH, gW, gD


def conv_project_cells(x: Tensor, w: "SplitConv", grid: Tuple[int, ...], ln=None) -> Tensor:
    lib = _lib.load()
    B, N, Cin = x.shape
    gH, gW, gD = _grid3(grid, N)
    f16 = _conv_f16(w)
    cells = conv_presplit(x, ln, f16)
    out = torch.empty((B, N, w.O), dtype=torch.float32, device=x.device)
    kH, kW, kD = w.kernel
    check(lib.haet_conv_project_cells_fwd(_p(cells), _p(w.f16() if f16 else w.lin.hi), _p(None if f16 else w.lin.lo), _p(w.lin.bias),
                                          B, gH, gW, gD, Cin, w.O, kH, kW, kD, _p(out), int(f16), _stream()),
          "haet_conv_project_cells_fwd")
    return out


def conv_project_cells_split(x: Tensor, w: "SplitConv", grid: Tuple[int, ...], p: SliceParams, H: int, C: int, ln=None) -> SplitPoints:
    lib = _lib.load()
    B, N, Cin = x.shape
    gH, gW, gD = _grid3(grid, N)
    f16 = _conv_f16(w)
    xc = conv_presplit(x, ln, f16)
    cells = torch.empty((B, N, w.O), dtype=torch.float32, device=x.device)
    s = torch.empty((B, N, H), dtype=torch.float32, device=x.device)
    kH, kW, kD = w.kernel
    check(lib.haet_conv_project_cells_split_fwd(_p(xc), _p(w.f16() if f16 else w.lin.hi), _p(None if f16 else w.lin.lo),
                                                _p(w.lin.bias), _p(p.wt), p.bt, p.base_temp, B, gH, gW, gD, Cin, w.O, kH, kW, kD,
                                                _p(cells), _p(s), int(f16), _stream()),
          "haet_conv_project_cells_split_fwd")
    return SplitPoints(cells, s, H, C)


def conv_project_supported(Cin: int, O: int) -> bool:
    return bool(_lib.load().haet_conv_project_supported(int(Cin), int(O)))


def conv_project(x: Tensor, w: SplitConv, grid: Tuple[int, ...], ln=None) -> Tensor:
    """x [B, N, C_in] channels-last over grid (H, W[, D]) -> [B, N, O]; same-padded convolution.
    `ln` = (weight, bias, eps): a LayerNorm over C_in applied first (only where conv_cells_applies: the pre-split pass)."""
    lib = _lib.load()
    x = _f32(x, "x")
    B, N, Cin = x.shape
    gH, gW = grid[0], grid[1]
    gD = grid[2] if len(grid) > 2 else 1
    if gH * gW * gD != N or Cin != w.Cin:
        raise HaetError("conv_project: x does not match the grid / weight")
    if ln is not None or conv_cells_applies(Cin, w.O, B * N, w.taps):
        return conv_project_cells(x, w, grid, ln)
    out = torch.empty((B, N, w.O), dtype=torch.float32, device=x.device)
    kH, kW, kD = w.kernel
    if _lib.precision() == "fp16x2" and w.f16_ok():
        check(lib.haet_conv_project_f16_fwd(_p(x), _p(w.f16()), _p(w.lin.bias), B, gH, gW, gD, Cin, w.O, kH, kW, kD, _p(out),
                                            _stream()), "haet_conv_project_f16_fwd")
        return out
    # small grids otherwise split the k-tiles over workgroups that add with float atomics (run-to-run order): under
    # torch.use_deterministic_algorithms(True) the fixed-order entry point is called -- a per-call choice; the library's
    # process-wide flag (haet_set_deterministic / HAET_DETERMINISTIC, read by the library itself) is never written here
    fn = "haet_conv_project_det_fwd" if torch.are_deterministic_algorithms_enabled() else "haet_conv_project_fwd"
    check(getattr(lib, fn)(_p(x), _p(w.lin.hi), _p(w.lin.lo), _p(w.lin.bias), B, gH, gW, gD, Cin, w.O,
                           kH, kW, kD, _p(out), _stream()), fn)
    return out
