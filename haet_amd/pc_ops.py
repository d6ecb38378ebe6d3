"""Torch-tensor front-ends of the stand-alone ErwinFlash entry points (include/haet_hip.h, last section): the ball tree
of raw point clouds, the radius graph, the MPNN stages and the streaming stages of the Erwin body.  Like ops.py these
only validate, allocate outputs and pass raw pointers to libhaet_hip.so on the current HIP stream; no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import HaetError, check
from .ops import SplitLinear, _f32, _p, _stream, linear

Tensor = torch.Tensor


def _i32_host(values: Sequence[int]):
    return (ctypes.c_int32 * len(values))(*[int(v) for v in values])


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


# --------------------------------------------------------------------------- tree
def balltree_big(pos: Tensor, cloud_sizes: Sequence[int], strides: Sequence[int], ball_sizes: Sequence[int], angle: float):
    """build_balltree_with_rotations (balltree.pyx:598-662) on the device for clouds of any size.
    Returns (tree_idx int32 [slots], tree_mask uint8 [slots], [rot_i int32 | None per layer])."""
    lib = _lib.load()
    pos = _f32(pos, "pos")
    N, d = pos.shape
    if sum(cloud_sizes) != N:
        raise HaetError(f"cloud sizes sum to {sum(cloud_sizes)}, positions have {N} rows")
    if len(strides) != len(ball_sizes) - 1:
        raise AssertionError("Strides must be one less than ball sizes")
    nc = len(cloud_sizes)
    sizes = _i32_host(cloud_sizes)
    slots = lib.haet_balltree_big_slots(sizes, nc)
    dev = pos.device
    tree_idx = torch.empty(slots, dtype=torch.int32, device=dev)
    tree_mask = torch.empty(slots, dtype=torch.uint8, device=dev)
    level_sizes, n = [], slots
    for i in range(len(ball_sizes)):
        level_sizes.append(n)
        if i < len(strides):
            if n % strides[i]:
                raise HaetError(f"stride {strides[i]} does not divide {n} tree slots")
            n //= strides[i]
    rot = torch.empty(sum(level_sizes), dtype=torch.int32, device=dev) if angle > 0 else None
    ws = torch.empty(lib.haet_balltree_big_workspace_bytes(sizes, nc, d), dtype=torch.uint8, device=dev)
    check(lib.haet_balltree_big_build(_p(pos), sizes, nc, d, len(ball_sizes), _i32_host(strides) if strides else None,
                                      _i32_host(ball_sizes), float(angle), _p(tree_idx), _p(tree_mask), _p(rot), _p(ws),
                                      ws.numel(), _stream()), "haet_balltree_big_build")
    rots: List[Optional[Tensor]] = []
    off = 0
    for n in level_sizes:
        rots.append(rot[off:off + n] if rot is not None else None)
        off += n
    return tree_idx, tree_mask, rots


# --------------------------------------------------------------------------- radius graph / MPNN
def radius_graph(pos: Tensor, r: float, batch_idx: Tensor, cloud_sizes: Sequence[int], max_num_neighbors: int = 32,
                 grid: Optional[bool] = None) -> Tensor:
    """torch_cluster.radius_graph(pos, r, batch=batch_idx, loop=True) -> edge_index int64 [2, E] (erwin_flash.py:160-166).
    grid=None: the uniform-grid search wherever it applies (max_num_neighbors <= 32), else all pairs of a cloud."""
    lib = _lib.load()
    pos = _f32(pos, "pos")
    N, d = pos.shape
    dev = pos.device
    nc = len(cloud_sizes)
    nbr = torch.empty((N, max_num_neighbors), dtype=torch.int32, device=dev)
    cnt = torch.empty(N, dtype=torch.int32, device=dev)
    if grid is None:
        grid = max_num_neighbors <= 32
    if grid:
        lo_hi = torch.stack([pos.min(dim=0)[0], pos.max(dim=0)[0]]).tolist()  # one host read (it sizes the cell table)
        cell = float(r) * (1.0 + 1.0 / 256.0)  # a little wider than r: the cell of a point is computed in fp32
        while True:
            G = [min(2048, int((lo_hi[1][k] - lo_hi[0][k]) / cell) + 1) if k < d else 1 for k in range(3)]
            if nc * G[0] * G[1] * G[2] <= (1 << 24) and all(G[k] < 2048 or (lo_hi[1][k] - lo_hi[0][k]) / cell < 2047 for k in range(d)):
                break
            cell *= 1.5
        mn = (ctypes.c_float * 3)(*[lo_hi[0][k] if k < d else 0.0 for k in range(3)])
        Gc = (ctypes.c_int32 * 3)(*G)
        cloud_of = batch_idx.to(torch.int32).contiguous()
        key = torch.empty(N, dtype=torch.int64, device=dev)
        check(lib.haet_pc_radius_cell_keys(_p(pos), _p(cloud_of), N, d, mn, cell, Gc, _p(key), _stream()), "haet_pc_radius_cell_keys")
        skey, order = torch.sort(key, stable=True)
        ncell = nc * G[0] * G[1] * G[2]
        cell_ptr = torch.searchsorted(skey, torch.arange(ncell + 1, device=dev, dtype=torch.int64)).to(torch.int32)
        check(lib.haet_pc_radius_neighbours_grid(_p(pos), _p(cloud_of), _p(order), _p(cell_ptr), N, d, mn, cell, Gc, float(r),
                                                 max_num_neighbors, _p(nbr), _p(cnt), _stream()), "haet_pc_radius_neighbours_grid")
    else:
        offs = torch.zeros(nc + 1, dtype=torch.int32, device=dev)
        offs[1:] = torch.tensor(list(cloud_sizes), dtype=torch.int32, device=dev).cumsum(0)
        b = batch_idx.to(torch.long)
        lo, hi = offs[b].contiguous(), offs[b + 1].contiguous()
        check(lib.haet_pc_radius_neighbours(_p(pos), _p(lo), _p(hi), N, d, float(r), max_num_neighbors, _p(nbr), _p(cnt), _stream()),
              "haet_pc_radius_neighbours")
    rowptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = cnt.to(torch.int64).cumsum(0)
    E = int(rowptr[-1].item())  # the edge list's size is data dependent: one host read, like the reference's own sync
    edge_index = torch.empty((2, E), dtype=torch.int64, device=dev)
    check(lib.haet_pc_radius_fill(_p(nbr), _p(cnt), _p(rowptr), N, max_num_neighbors, E, _p(edge_index), _stream()),
          "haet_pc_radius_fill")
    return edge_index


def edges_to_csr(edge_index: Tensor, N: int):
    """(rowptr int32 [N+1], src int32 [E]): the senders of every receiver's edges (receiver = edge_index[1])."""
    row, col = edge_index[0].to(torch.long), edge_index[1].to(torch.long)
    if row.numel():
        # the indices address rows of device tensors inside the message kernels (and atomic writes in their backward):
        # one host read (the edge list's producer, radius_graph, synchronises anyway)
        lo, hi = torch.aminmax(edge_index)
        if int(lo) < 0 or int(hi) >= N:
            raise HaetError(f"edge_index holds node indices outside [0, {N}): min {int(lo)}, max {int(hi)}")
    if col.numel() > 1 and not bool((col[1:] >= col[:-1]).all()):
        col, order = torch.sort(col, stable=True)
        row = row[order]
    rowptr = torch.zeros(N + 1, dtype=torch.int32, device=edge_index.device)
    rowptr[1:] = torch.bincount(col, minlength=N).cumsum(0).to(torch.int32)
    return rowptr, row.to(torch.int32).contiguous()


def mpnn_message(AB: Tensor, pos: Tensor, rowptr: Tensor, src: Tensor, Wc: Tensor, bias: Tensor, c: int, out: Tensor, eps: float = 1e-5):
    lib = _lib.load()
    N = pos.shape[0]
    check(lib.haet_pc_mpnn_message(_p(AB), AB.stride(0), _p(pos), pos.shape[1], _p(rowptr), _p(src), _p(Wc), _p(bias), N, c, eps,
                                   _p(out), out.stride(0), _stream()), "haet_pc_mpnn_message")


def ln_residual(u: Tensor, h: Tensor, h2: Optional[Tensor], c: int, eps: float = 1e-5):
    lib = _lib.load()
    check(lib.haet_pc_ln_residual(_p(u), u.stride(0), h.shape[0], c, eps, _p(h), h.stride(0), _p(h2), 0 if h2 is None else h2.stride(0),
                                  _stream()), "haet_pc_ln_residual")


def ln_bwd(v: Tensor, dU: Tensor, c: int, eps: float = 1e-5) -> Tensor:
    """Gradient of LayerNorm without affine w.r.t. its input v [rows, >= c], from dU [rows, >= c]."""
    lib = _lib.load()
    dV = torch.empty((v.shape[0], c), dtype=torch.float32, device=v.device)
    check(lib.haet_pc_ln_bwd(_p(v), v.stride(0), _p(dU), dU.stride(0), v.shape[0], c, eps, _p(dV), c, _stream()), "haet_pc_ln_bwd")
    return dV


def mpnn_message_bwd(AB: Tensor, pos: Tensor, rowptr: Tensor, src: Tensor, Wc: Tensor, bias: Tensor, dagg: Tensor, c: int,
                     eps: float = 1e-5):
    """Backward of mpnn_message: (dAB [N, 2c] = [dA | dB], dWc [c, d]) from dagg [N, c] (any row stride)."""
    lib = _lib.load()
    N, d = pos.shape
    dAB = torch.zeros((N, 2 * c), dtype=torch.float32, device=AB.device)
    part = torch.zeros((lib.haet_pc_mpnn_bwd_waves(N), c, 3), dtype=torch.float32, device=AB.device)
    check(lib.haet_pc_mpnn_message_bwd(_p(AB), AB.stride(0), _p(pos), d, _p(rowptr), _p(src), _p(Wc), _p(bias), _p(dagg), dagg.stride(0),
                                       N, c, eps, _p(dAB), 2 * c, _p(part), _stream()), "haet_pc_mpnn_message_bwd")
    return dAB, part.sum(dim=0)[:, :d]


# --------------------------------------------------------------------------- Erwin body stages
def gather_rows(x: Tensor, idx: Tensor, c: int) -> Tensor:
    lib = _lib.load()
    out = torch.empty((idx.numel(), c), dtype=torch.float32, device=x.device)
    check(lib.haet_pc_gather_rows(_p(x), x.stride(0), _p(idx), idx.numel(), c, _p(out), c, _stream()), "haet_pc_gather_rows")
    return out


def scatter_masked(X: Tensor, tree_idx: Tensor, mask: Tensor, c: int, n_out: int) -> Tensor:
    lib = _lib.load()
    out = torch.empty((n_out, c), dtype=torch.float32, device=X.device)
    check(lib.haet_pc_scatter_masked(_p(X), X.stride(0), _p(tree_idx), _p(mask), X.shape[0], c, _p(out), c, _stream()),
          "haet_pc_scatter_masked")
    return out


def rows_add(X: Tensor, perm: Optional[Tensor], R: Tensor, c: int):
    lib = _lib.load()
    check(lib.haet_pc_rows_add(_p(X), X.stride(0), _p(perm), _p(R), R.stride(0), X.shape[0], c, _stream()), "haet_pc_rows_add")


def ball_rel(P: Tensor, perm: Optional[Tensor], ball: int) -> Tensor:
    lib = _lib.load()
    rel = torch.empty_like(P)
    check(lib.haet_pc_ball_rel(_p(P), _p(perm), P.shape[0], ball, P.shape[1], _p(rel), _stream()), "haet_pc_ball_rel")
    return rel


def rms_pe(X: Tensor, perm: Optional[Tensor], rel: Optional[Tensor], w: Tensor, pe_w: Optional[Tensor], pe_b: Optional[Tensor], c: int) -> Tensor:
    lib = _lib.load()
    Y = torch.empty((X.shape[0], c), dtype=torch.float32, device=X.device)
    d = 0 if rel is None else rel.shape[1]
    check(lib.haet_pc_rms_pe(_p(X), X.stride(0), _p(perm), _p(rel), X.shape[0], c, d, _p(w), _p(pe_w), _p(pe_b), _p(Y), c, _stream()),
          "haet_pc_rms_pe")
    return Y


def ball_attention_supported(c: int, heads: int, ball: int) -> bool:
    return bool(_lib.load().haet_pc_ball_attention_supported(int(c), int(heads), int(ball)))


def ball_attention(qkv: Tensor, c: int, heads: int, ball: int) -> Tensor:
    lib = _lib.load()
    out = torch.empty((qkv.shape[0], c), dtype=torch.float32, device=qkv.device)
    check(lib.haet_pc_ball_attention(_p(qkv), qkv.stride(0), qkv.shape[0], c, heads, ball, _p(out), c, _stream()),
          "haet_pc_ball_attention")
    return out


def rms_bwd_(X: Tensor, perm: Optional[Tensor], dY: Tensor, w: Tensor, c: int, dX: Tensor, accumulate: bool = True):
    """dX[slot(p)] (+)= gradient of RMSNorm(X[slot(p)]) * w w.r.t. X, from dY[p] (in place on dX)."""
    lib = _lib.load()
    check(lib.haet_pc_rms_bwd(_p(X), X.stride(0), _p(perm), _p(dY), dY.stride(0), _p(w), X.shape[0], c, _p(dX), dX.stride(0),
                              1 if accumulate else 0, _stream()), "haet_pc_rms_bwd")


def swiglu_bwd(U: Tensor, dH: Tensor, hid: int) -> Tensor:
    lib = _lib.load()
    dU = torch.empty((U.shape[0], 2 * hid), dtype=torch.float32, device=U.device)
    check(lib.haet_pc_swiglu_bwd(_p(U), U.stride(0), _p(dH), dH.stride(0), U.shape[0], hid, _p(dU), 2 * hid, _stream()),
          "haet_pc_swiglu_bwd")
    return dU


def ball_attention_bwd_supported(c: int, heads: int, ball: int) -> bool:
    return bool(_lib.load().haet_pc_ball_attention_bwd_supported(int(c), int(heads), int(ball)))


def ball_attention_bwd(qkv: Tensor, dO: Tensor, c: int, heads: int, ball: int) -> Tensor:
    lib = _lib.load()
    dqkv = torch.empty((qkv.shape[0], 3 * c), dtype=torch.float32, device=qkv.device)
    check(lib.haet_pc_ball_attention_bwd(_p(qkv), qkv.stride(0), _p(dO), dO.stride(0), qkv.shape[0], c, heads, ball, _p(dqkv), 3 * c,
                                         _stream()), "haet_pc_ball_attention_bwd")
    return dqkv


def swiglu(U: Tensor, hid: int) -> Tensor:
    lib = _lib.load()
    H = torch.empty((U.shape[0], hid), dtype=torch.float32, device=U.device)
    check(lib.haet_pc_swiglu(_p(U), U.stride(0), U.shape[0], hid, _p(H), hid, _stream()), "haet_pc_swiglu")
    return H


def pool_gather(X: Tensor, P: Tensor, c: int, st: int, ldy: int):
    lib = _lib.load()
    rows = X.shape[0] // st
    Y = torch.empty((rows, ldy), dtype=torch.float32, device=X.device)
    Pc = torch.empty((rows, P.shape[1]), dtype=torch.float32, device=X.device)
    check(lib.haet_pc_pool_gather(_p(X), X.stride(0), _p(P), rows, c, P.shape[1], st, _p(Y), ldy, _p(Pc), _stream()), "haet_pc_pool_gather")
    return Y, Pc


def unpool_input(Xp: Tensor, Pp: Tensor, Pchild: Tensor, c: int, st: int, ldy: int) -> Tensor:
    lib = _lib.load()
    Y = torch.empty((Xp.shape[0], ldy), dtype=torch.float32, device=Xp.device)
    check(lib.haet_pc_unpool_input(_p(Xp), Xp.stride(0), _p(Pp), _p(Pchild), Xp.shape[0], c, Pp.shape[1], st, _p(Y), ldy, _stream()),
          "haet_pc_unpool_input")
    return Y


def unpool_merge(skip: Tensor, Yp: Tensor, c: int, st: int) -> Tensor:
    lib = _lib.load()
    out = torch.empty((skip.shape[0], c), dtype=torch.float32, device=skip.device)
    check(lib.haet_pc_unpool_merge(_p(skip), skip.stride(0), _p(Yp), Yp.stride(0), skip.shape[0], c, st, _p(out), c, _stream()),
          "haet_pc_unpool_merge")
    return out


def batch_norm_(X: Tensor, c: int, bn: torch.nn.BatchNorm1d, training: bool):
    """nn.BatchNorm1d in place over the rows of X (batch statistics + running-statistics update when training)."""
    lib = _lib.load()
    rows = X.shape[0]
    if training:
        part = torch.empty((lib.haet_pc_bn_blocks(rows), 2, c), dtype=torch.float32, device=X.device)
        stats = torch.empty((2, c), dtype=torch.float32, device=X.device)
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        track = bn.track_running_stats and bn.running_mean is not None
        check(lib.haet_pc_bn_stats(_p(X), X.stride(0), rows, c, mom, _p(part), _p(stats),
                                   _p(bn.running_mean if track else None), _p(bn.running_var if track else None),
                                   _p(bn.num_batches_tracked if track else None), _stream()), "haet_pc_bn_stats")
        mean, var = stats[0], stats[1]
    else:
        mean, var = bn.running_mean, bn.running_var
    check(lib.haet_pc_bn_apply(_p(X), X.stride(0), rows, c, _p(mean), _p(var), float(bn.eps), _p(bn.weight.detach()), _p(bn.bias.detach()),
                               _stream()), "haet_pc_bn_apply")


class SwigluLinear:
    """w1 | w2 of a SwiGLU as ONE GEMM with the gate in its epilogue (haet_linear_swiglu_fwd): rows interleaved."""

    def __init__(self, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor):
        hid, K = w1.shape
        self.hid, self.K = hid, K
        W = torch.stack([w1.detach(), w2.detach()], dim=1).reshape(2 * hid, K)
        b = torch.stack([b1.detach(), b2.detach()], dim=1).reshape(2 * hid)
        self.w = SplitLinear(W, b)

    @staticmethod
    def supported(K: int, hid: int, rows: int) -> bool:
        return bool(_lib.load().haet_linear_swiglu_supported(int(K), int(2 * hid), int(rows)))

    def __call__(self, x: Tensor, rms_weight: Optional[Tensor] = None) -> Tensor:
        """rms_weight: nn.RMSNorm(K) (eps = machine epsilon, layers.py:25-26) is applied to x's rows inside the GEMM's loader."""
        lib = _lib.load()
        H = torch.empty((x.shape[0], self.hid), dtype=torch.float32, device=x.device)
        check(lib.haet_linear_swiglu_fwd(_p(x), _p(rms_weight), 1.1920928955078125e-07, _p(self.w.hi), _p(self.w.lo), _p(self.w.bias),
                                         x.shape[0], self.K, 2 * self.hid, _p(H), _stream()), "haet_linear_swiglu_fwd")
        return H


class PaddedLinear:
    """A Linear [O, K] for the split-bf16 GEMM (haet_linear_fwd: K % 32 = 0, O % 128 = 0): zero-padded to that envelope
    once per weight version; inputs carry Kp columns (zeros past K), outputs Op columns (zeros past O)."""

    def __init__(self, weight: Tensor, bias: Optional[Tensor]):
        O, K = weight.shape
        self.O, self.K, self.Op, self.Kp = O, K, round_up(O, 128), round_up(K, 32)
        if self.Op == O and self.Kp == K:  # already inside the envelope: no padded copies (rebuilt every optimizer step)
            self.w = SplitLinear(weight, bias)
            return
        W = torch.zeros((self.Op, self.Kp), dtype=torch.float32, device=weight.device)
        W[:O, :K] = weight.detach()
        b = torch.zeros(self.Op, dtype=torch.float32, device=weight.device)
        if bias is not None:
            b[:O] = bias.detach()
        self.w = SplitLinear(W, b)

    def __call__(self, x: Tensor) -> Tensor:
        if x.shape[-1] != self.Kp or not x.is_contiguous():
            raise HaetError(f"PaddedLinear: input width {tuple(x.shape)} != padded K {self.Kp}")
        return linear(x, self.w)

    def compact(self, y: Tensor) -> Tensor:
        return y if self.Op == self.O else y[:, :self.O].contiguous()
