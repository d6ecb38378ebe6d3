"""ErwinTransformer.forward on RAW point clouds -- host side.

What the reference does here (models/components/erwinflash/erwin_flash.py:138-195 with embed=True / mp_steps>0;
benchmarks/02.1-Car-Design-ShapeNetCar/main.py:50-99): ball tree + rotated re-partitions on the HOST, radius graph,
Linear embedding + MPNN steps, gather into tree order, encoder / bottleneck / decoder of ball-attention blocks with
BallPooling / BallUnpooling in between, scatter back to the input order.  Here every one of those stages is a HIP
launch on the current stream (haet_amd/pc_ops.py -> csrc/balltree_big.hip, csrc/pointcloud.hip, and the split-bf16
GEMMs of csrc/gemm_x3.hip for the Linears); the state stays in HBM as [tree slots][channels] fp32.

Gradients (`forward_autograd`): the transformer blocks -- nearly all of the arithmetic -- run forward AND backward on the HIP
stages (`ErwinBlockFunction`: input kept, block recomputed in the backward, input gradients through the split-bf16 GEMM
on the transposed weights, RMSNorm / SwiGLU / ball attention through their backward kernels, weight gradients as fp32
library GEMMs); embedding / MPNN, pooling, BatchNorm and the gathers differentiate through torch device ops on the
HIP-built tree and radius graph.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn.functional as F

from ... import _lib, ops, pc_ops
from ..._lib import HaetError

Tensor = torch.Tensor


def supported(erwin) -> Optional[str]:
    """None when the HIP point-cloud path covers this configuration, else the reason."""
    if erwin.dimensionality not in (2, 3):
        return f"dimensionality {erwin.dimensionality}"
    layers = list(erwin.encoder) + [erwin.bottleneck] + (list(erwin.decoder) if erwin.decode else [])
    for L in layers:
        c = L.hidden_dim
        if c % 32 or c > 1024 or c * L.mlp_ratio > 1024:
            return f"width {c} (multiple of 32, hidden <= 1024)"
        if not pc_ops.ball_attention_supported(c, L.num_heads, L.ball_size):
            return f"ball attention c={c} heads={L.num_heads} ball={L.ball_size}"
    return None


class _Weights:
    """Split-bf16 operands of every Linear, rebuilt when a parameter changes (same key as the packed token path)."""

    def __init__(self, erwin):
        P = pc_ops.PaddedLinear
        self.blocks = {}
        layers = list(erwin.encoder) + [erwin.bottleneck] + (list(erwin.decoder) if erwin.decode else [])
        for L in layers:
            for blk in L.blocks:
                a, s = blk.BMSA, blk.swiglu
                w12 = torch.cat([s.w1.weight, s.w2.weight], 0)
                self.blocks[id(blk)] = dict(
                    qkv=P(a.qkv.weight, a.qkv.bias), proj=P(a.proj.weight, a.proj.bias),
                    w12=P(w12, torch.cat([s.w1.bias, s.w2.bias], 0)), w3=P(s.w3.weight, s.w3.bias),
                    # the same matrices transposed: input gradients of the Linears (training)
                    qkvT=P(a.qkv.weight.t(), None), projT=P(a.proj.weight.t(), None), w12T=P(w12.t(), None),
                    w3T=P(s.w3.weight.t(), None),
                    # forward only: w1 | w2 with the SwiGLU gate in the GEMM's epilogue (the [rows, 2 hid] tensor never exists)
                    glu=(pc_ops.SwigluLinear(s.w1.weight, s.w1.bias, s.w2.weight, s.w2.bias)
                         if pc_ops.SwigluLinear.supported(s.w1.weight.shape[1], s.w1.weight.shape[0], 64) else None),
                    norm1=blk.norm1.weight.detach().float().contiguous(), norm2=blk.norm2.weight.detach().float().contiguous(),
                    pe_w=a.pe_proj.weight.detach().float().contiguous(), pe_b=a.pe_proj.bias.detach().float().contiguous())
            for name in ("pool", "unpool"):
                m = getattr(L, name, None)
                if m is not None:
                    self.blocks[id(m)] = P(m.proj.weight, m.proj.bias)
        if erwin.use_embedding:
            e = erwin.embed
            self.embed = P(e.embed_fn.weight, e.embed_fn.bias) if not ops.linear_narrow_supported(*e.embed_fn.weight.shape[::-1]) else None
            c = e.embed_fn.weight.shape[0]
            self.mp = []
            for msg, upd in zip(e.mpnn.fused_message_update, e.mpnn.update_fns):
                W = msg[0].weight
                wab = torch.cat([W[:, :c], W[:, c:2 * c]], 0)
                self.mp.append(dict(ab=P(wab, None), wc=W[:, 2 * c:].detach().float().contiguous(),
                                    b=msg[0].bias.detach().float().contiguous(), upd=P(upd[0].weight, upd[0].bias),
                                    abT=P(wab.t(), None), updT=P(upd[0].weight.t(), None)))


def _weights(erwin) -> _Weights:
    key = erwin._pack_key()
    hit = erwin.__dict__.get("_pc_weights")
    if hit is None or hit[0] != key:
        hit = (key, _Weights(erwin))
        erwin.__dict__["_pc_weights"] = hit
    return hit[1]


def cloud_sizes(batch_idx: Tensor) -> List[int]:
    """Points per cloud (batch_idx is sorted, contiguous from 0 like balltree.pyx:459-470 requires); one host read."""
    b = batch_idx.to(torch.long)
    if b.numel() > 1 and not bool((b[1:] >= b[:-1]).all()):
        raise HaetError("batch_idx must be sorted (the points of a cloud contiguous), as the reference's tree builder requires")
    sizes = torch.bincount(b).tolist()
    if sizes and min(sizes) < 2:
        raise HaetError(f"every cloud needs at least 2 points (cloud sizes {sizes[:8]}...)")
    return sizes


def build_graph(erwin, pos: Tensor, batch_idx: Tensor, sizes, edge_index: Optional[Tensor], radius: Optional[float]):
    if edge_index is None and erwin.use_embedding and erwin.mp_steps > 0:
        if radius is None:
            raise AssertionError("radius (float) must be provided if edge_index is not given to build radius graph")
        edge_index = pc_ops.radius_graph(pos, radius, batch_idx, sizes)
    return edge_index


def build_tree(erwin, pos: Tensor, sizes, side: Optional[torch.cuda.Stream] = None):
    """The ball tree and its rotated re-partitions.  `side`: build on that stream -- the builder is a chain of small
    launches with host waits in between (a few workgroups each), which the embedding's kernels on the caller's stream
    overlap; the caller's stream then waits for the side stream."""
    if side is None:
        return pc_ops.balltree_big(pos, sizes, erwin.strides, erwin.ball_sizes, float(erwin.rotate))
    cur = torch.cuda.current_stream(pos.device)
    side.wait_stream(cur)  # (pos may still be in flight)
    with torch.cuda.stream(side):
        tree_idx, tree_mask, rots = pc_ops.balltree_big(pos, sizes, erwin.strides, erwin.ball_sizes, float(erwin.rotate))
    cur.wait_stream(side)
    for t in [tree_idx, tree_mask] + [r for r in rots if r is not None]:
        t.record_stream(cur)
    return tree_idx, tree_mask, rots


def build_structures(erwin, pos: Tensor, batch_idx: Tensor, edge_index: Optional[Tensor], radius: Optional[float]):
    sizes = cloud_sizes(batch_idx)
    tree_idx, tree_mask, rots = build_tree(erwin, pos, sizes)
    return sizes, tree_idx, tree_mask, rots, build_graph(erwin, pos, batch_idx, sizes, edge_index, radius)


# --------------------------------------------------------------------------- HIP forward
def _embed(erwin, W: _Weights, feat: Tensor, pos: Tensor, edge_index: Optional[Tensor]) -> Tensor:
    e = erwin.embed
    c = e.embed_fn.weight.shape[0]
    if W.embed is None:
        h = ops.linear_narrow(feat, e.embed_fn.weight, e.embed_fn.bias)
    else:
        h = W.embed.compact(W.embed(F.pad(feat, (0, W.embed.Kp - feat.shape[1]))))
    if erwin.mp_steps == 0:
        return h
    N = feat.shape[0]
    rowptr, src = pc_ops.edges_to_csr(edge_index, N)
    hcat = torch.empty((N, 2 * c), dtype=torch.float32, device=feat.device)
    hcat[:, :c] = h
    for st in W.mp:
        AB = st["ab"](h)
        pc_ops.mpnn_message(AB, pos, rowptr, src, st["wc"], st["b"], c, hcat[:, c:])
        u = st["upd"](hcat)
        pc_ops.ln_residual(u, h, hcat, c)
    return h


def _block(X: Tensor, P: Tensor, perm: Optional[Tensor], blk, w: dict, d: int):
    c, ball, heads = X.shape[1], blk.ball_size, blk.BMSA.num_heads
    rel = pc_ops.ball_rel(P, perm, ball)
    Y = pc_ops.rms_pe(X, perm, rel, w["norm1"], w["pe_w"], w["pe_b"], c)
    A = pc_ops.ball_attention(w["qkv"](Y), c, heads, ball)
    pc_ops.rows_add(X, perm, w["proj"](A), c)
    if w["glu"] is not None and X.shape[0] >= 64:  # RMSNorm in the GEMM's loader, SwiGLU gate in its epilogue
        H = w["glu"](X, w["norm2"])
    else:
        Y = pc_ops.rms_pe(X, None, None, w["norm2"], None, None, c)
        H = pc_ops.swiglu(w["w12"](Y), w["w3"].K)
    pc_ops.rows_add(X, None, w["w3"](H), c)


def _layer(X, P, rot, L, W: _Weights, d):
    use_rot = any(L.rotate) and rot is not None
    for j, blk in enumerate(L.blocks):
        _block(X, P, rot if (use_rot and L.rotate[j]) else None, blk, W.blocks[id(blk)], d)


def forward_hip(erwin, feat: Tensor, pos: Tensor, batch_idx: Tensor, edge_index: Optional[Tensor], radius: Optional[float]):
    why = supported(erwin)
    if why is not None:
        raise HaetError(f"stand-alone ErwinFlash outside the HIP point-cloud path: {why}")
    feat, pos = ops._f32(feat, "node_features"), ops._f32(pos, "node_positions")
    d = erwin.dimensionality
    W = _weights(erwin)
    sizes = cloud_sizes(batch_idx)
    # graph + embedding are enqueued first; the tree is built beside them on a side stream (they meet at the gather)
    edge_index = build_graph(erwin, pos, batch_idx, sizes, edge_index, radius)
    x = _embed(erwin, W, feat, pos, edge_index) if erwin.use_embedding else feat
    side = None
    if erwin.use_embedding and not os.environ.get("HAET_PC_ONE_STREAM"):
        side = erwin.__dict__.get("_pc_side_stream")
        if side is None or side.device != pos.device:
            side = erwin.__dict__["_pc_side_stream"] = torch.cuda.Stream(device=pos.device)
    tree_idx, tree_mask, rots = build_tree(erwin, pos, sizes, side)
    c0 = x.shape[1]
    X = pc_ops.gather_rows(x, tree_idx, c0)
    P = pos.index_select(0, tree_idx.to(torch.long))
    train = erwin.training
    stack = []
    for i, L in enumerate(erwin.encoder):
        _layer(X, P, rots[i], L, W, d)
        stack.append((X, P, rots[i]))
        st = L.pool.stride
        if st != 1:  # (stride 1: the node passes through, pooling.py:26-27)
            pl = W.blocks[id(L.pool)]
            Y, Pc = pc_ops.pool_gather(X, P, X.shape[1], st, pl.Kp)
            X = pl.compact(pl(Y))
            pc_ops.batch_norm_(X, X.shape[1], L.pool.norm, train)
            P = Pc
        else:
            X = X.clone()
    _layer(X, P, rots[len(erwin.encoder)], erwin.bottleneck, W, d)
    if not erwin.decode:
        b = batch_idx[tree_idx.to(torch.long)]
        for s in erwin.strides:
            b = b[::s]
        return X, b
    for k, L in enumerate(erwin.decoder):
        cx, cp, crot = stack[len(erwin.encoder) - 1 - k]
        st = L.unpool.stride
        ul = W.blocks[id(L.unpool)]
        Y = pc_ops.unpool_input(X, P, cp, X.shape[1], st, ul.Kp)
        X = pc_ops.unpool_merge(cx, ul(Y), cx.shape[1], st)
        pc_ops.batch_norm_(X, X.shape[1], L.unpool.norm, train)
        P = cp
        _layer(X, P, crot, L, W, d)
    return pc_ops.scatter_masked(X, tree_idx, tree_mask, X.shape[1], feat.shape[0])


# --------------------------------------------------------------------------- training: blocks on the HIP kernels
class ErwinBlockFunction(torch.autograd.Function):
    """One ErwinTransformerBlock (layers.py:30-32) over the tree-ordered state, forward AND backward on the HIP stages:
    the forward keeps only its input, the backward recomputes the block (11 launches) and walks it back -- input
    gradients of the four Linears through the split-bf16 GEMM on the transposed weights, RMSNorm / SwiGLU / ball
    attention through their backward kernels (csrc/pointcloud.hip); weight gradients are fp32 library GEMMs over the
    rows.  `perm` = the block's rotated order (int32) or None."""

    @staticmethod
    def forward(ctx, X, P, perm, blk, w, *params):
        X = ops._f32(X, "x")
        out = X.clone()
        _block(out, P, perm, blk, w, P.shape[1])
        ctx.save_for_backward(X, P, perm if perm is not None else torch.empty(0, dtype=torch.int32, device=X.device))
        ctx.blk, ctx.w = blk, w
        return out

    @staticmethod
    def backward(ctx, g):
        X1, P, perm = ctx.saved_tensors
        perm = perm if perm.numel() else None
        blk, w = ctx.blk, ctx.w
        c, ball, heads, hid = X1.shape[1], blk.ball_size, blk.BMSA.num_heads, w["w3"].K
        g = ops._f32(g, "grad")
        # the block again (nothing but its input was kept)
        rel = pc_ops.ball_rel(P, perm, ball)
        Y1 = pc_ops.rms_pe(X1, perm, rel, w["norm1"], w["pe_w"], w["pe_b"], c)
        QKV = w["qkv"](Y1)
        A = pc_ops.ball_attention(QKV, c, heads, ball)
        X2 = X1.clone()
        pc_ops.rows_add(X2, perm, w["proj"](A), c)
        Y2 = pc_ops.rms_pe(X2, None, None, w["norm2"], None, None, c)
        U = w["w12"](Y2)
        H = pc_ops.swiglu(U, hid)
        # x_out = x2 + w3(H)
        dH = w["w3T"](g)
        dU = pc_ops.swiglu_bwd(U, dH, hid)
        dY2 = w["w12T"](dU)
        dX = g.clone()
        pc_ops.rms_bwd_(X2, None, dY2, w["norm2"], c, dX)
        xh2 = X2 * torch.rsqrt(X2.pow(2).mean(-1, keepdim=True) + torch.finfo(torch.float32).eps)
        g_norm2 = (dY2[:, :c] * xh2).sum(0)
        g_w12, g_b12 = ops.linear_wgrad(dU, Y2)  # (split-bf16 G^T X kernel where both widths are multiples of 128)
        g_w3, g_b3 = ops.linear_wgrad(g, H)
        del U, H, dH, Y2, xh2
        # x2 = x1 + scatter(proj(attention(qkv(y1))))
        dR = pc_ops.gather_rows(dX, perm, c) if perm is not None else dX
        dA = w["projT"](dR)
        g_wp, g_bp = ops.linear_wgrad(dR, A)
        dQKV = pc_ops.ball_attention_bwd(QKV, dA, c, heads, ball)
        dY1 = w["qkvT"](dQKV)
        g_wq, g_bq = ops.linear_wgrad(dQKV, Y1)
        pc_ops.rms_bwd_(X1, perm, dY1, w["norm1"], c, dX)
        Xs = X1.index_select(0, perm.long()) if perm is not None else X1
        xh1 = Xs * torch.rsqrt(Xs.pow(2).mean(-1, keepdim=True) + torch.finfo(torch.float32).eps)
        d1 = dY1[:, :c]
        # (d columns of [c, n] @ [n, d]: three fused multiply-reduce passes; the library GEMM for this shape takes 0.65 ms)
        g_pew = torch.stack([(d1 * rel[:, k:k + 1]).sum(0) for k in range(rel.shape[1])], dim=1)
        g_norm1, g_peb = (d1 * xh1).sum(0), d1.sum(0)
        return (dX, None, None, None, None, g_norm1, g_pew, g_peb, g_wq, g_bq, g_wp, g_bp, g_norm2,
                g_w12[:hid], g_b12[:hid], g_w12[hid:2 * hid], g_b12[hid:2 * hid], g_w3, g_b3)


class MpnnStepFunction(torch.autograd.Function):
    """One message-passing step (mpnn.py:48-68) forward and backward on the HIP stages: node-level GEMMs for the h_j / h_i
    parts of the message Linear, the per-edge kernel for GELU + LayerNorm + mean, and their backward (sender gradients
    by float atomics, the position columns' gradient from per-wave partial sums).  Nothing of size [edges, .] exists."""

    @staticmethod
    def forward(ctx, h, pos, rowptr, src, st, w_msg, b_msg, w_upd, b_upd):
        h = ops._f32(h, "h")
        c = h.shape[1]
        AB = st["ab"](h)
        hcat = torch.empty((h.shape[0], 2 * c), dtype=torch.float32, device=h.device)
        hcat[:, :c] = h
        pc_ops.mpnn_message(AB, pos, rowptr, src, st["wc"], st["b"], c, hcat[:, c:])
        v = st["upd"](hcat)
        out = h.clone()
        pc_ops.ln_residual(v, out, None, c)
        ctx.save_for_backward(pos, rowptr, src, AB, hcat, v)
        ctx.st = st
        return out

    @staticmethod
    def backward(ctx, g):
        pos, rowptr, src, AB, hcat, v = ctx.saved_tensors
        st = ctx.st
        c, d = hcat.shape[1] // 2, pos.shape[1]
        g = ops._f32(g, "grad")
        dv = pc_ops.ln_bwd(v, g, c)
        g_wu, g_bu = ops.linear_wgrad(dv, hcat)
        dhcat = st["updT"](dv)
        dAB, g_wc = pc_ops.mpnn_message_bwd(AB, pos, rowptr, src, st["wc"], st["b"], dhcat[:, c:2 * c], c)
        g_wab, _ = ops.linear_wgrad(dAB, hcat[:, :c].contiguous())
        dh = g + dhcat[:, :c] + st["abT"](dAB)[:, :c]
        g_wm = torch.cat([g_wab[:c], g_wab[c:2 * c], g_wc], dim=1)
        return dh, None, None, None, None, g_wm, dAB[:, c:].sum(0), g_wu, g_bu


def _block_params(blk):
    a, s = blk.BMSA, blk.swiglu
    return (blk.norm1.weight, a.pe_proj.weight, a.pe_proj.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
            blk.norm2.weight, s.w1.weight, s.w1.bias, s.w2.weight, s.w2.bias, s.w3.weight, s.w3.bias)


def hip_block_training_supported(erwin) -> bool:
    if supported(erwin) is not None:
        return False
    layers = list(erwin.encoder) + [erwin.bottleneck] + (list(erwin.decoder) if erwin.decode else [])
    return all(pc_ops.ball_attention_bwd_supported(L.hidden_dim, L.num_heads, L.ball_size) for L in layers)


# --------------------------------------------------------------------------- differentiable restatement (training)
def _rms(x, w):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + torch.finfo(x.dtype).eps) * w


def _block_autograd(x, pos, blk):
    c, ball, heads = x.shape[1], blk.ball_size, blk.BMSA.num_heads
    a, s = blk.BMSA, blk.swiglu
    with torch.no_grad():
        p = pos.view(-1, ball, pos.shape[1])
        rel = (p - p.mean(1, keepdim=True)).reshape(-1, pos.shape[1])
    y = _rms(x, blk.norm1.weight) + F.linear(rel, a.pe_proj.weight, a.pe_proj.bias)
    qkv = F.linear(y, a.qkv.weight, a.qkv.bias).view(-1, ball, 3, heads, c // heads)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    o = F.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(-1, c)
    x = x + F.linear(o, a.proj.weight, a.proj.bias)
    y = _rms(x, blk.norm2.weight)
    return x + F.linear(F.linear(y, s.w2.weight, s.w2.bias) * F.silu(F.linear(y, s.w1.weight, s.w1.bias)), s.w3.weight, s.w3.bias)


def _layer_autograd(x, pos, rot, L, W=None):
    """W (the split-bf16 operands): the blocks run as ErwinBlockFunction on the HIP stages; None: torch device ops."""
    use_rot = any(L.rotate) and rot is not None
    if W is not None:
        rot32 = rot.to(torch.int32) if use_rot else None
        for j, blk in enumerate(L.blocks):
            x = ErwinBlockFunction.apply(x, pos, rot32 if (use_rot and L.rotate[j]) else None, blk, W.blocks[id(blk)],
                                         *_block_params(blk))
        return x
    inv = torch.argsort(rot) if use_rot else None
    for j, blk in enumerate(L.blocks):
        if use_rot and L.rotate[j]:
            x = _block_autograd(x[rot], pos[rot], blk)[inv]
        else:
            x = _block_autograd(x, pos, blk)
    return x


def forward_autograd(erwin, feat, pos, batch_idx, edge_index, radius, hip_blocks: Optional[bool] = None):
    """Training forward.  hip_blocks (default: wherever the configuration allows, HAET_PC_TORCH_BLOCKS=1 disables): the
    transformer blocks -- where nearly all of the arithmetic is -- run forward and backward on the HIP stages
    (ErwinBlockFunction); embedding / MPNN, pooling, BatchNorm and the gathers differentiate through torch device ops."""
    pos = pos.detach().float().contiguous()
    d = erwin.dimensionality
    if hip_blocks is None:
        hip_blocks = not os.environ.get("HAET_PC_TORCH_BLOCKS") and hip_block_training_supported(erwin)
    W = _weights(erwin) if hip_blocks else None
    sizes, tree_idx, tree_mask, rots, edge_index = build_structures(erwin, pos, batch_idx, edge_index, radius)
    tree_idx = tree_idx.to(torch.long)
    rots = [None if r is None else r.to(torch.long) for r in rots]
    x = feat
    if erwin.use_embedding:
        e = erwin.embed
        x = F.linear(feat, e.embed_fn.weight, e.embed_fn.bias)
        if erwin.mp_steps > 0 and W is not None and x.shape[1] % 128 == 0 and x.shape[1] <= 512:
            rowptr, src = pc_ops.edges_to_csr(edge_index, x.shape[0])
            for st, msg, upd in zip(W.mp, e.mpnn.fused_message_update, e.mpnn.update_fns):
                x = MpnnStepFunction.apply(x, pos, rowptr, src, st, msg[0].weight, msg[0].bias, upd[0].weight, upd[0].bias)
        elif erwin.mp_steps > 0:
            row, col = edge_index[0].to(torch.long), edge_index[1].to(torch.long)
            attr = pos[row] - pos[col]
            c = x.shape[1]
            cnt = torch.zeros(x.shape[0], device=x.device).index_add_(0, col, torch.ones(col.shape[0], device=x.device)).clamp(min=1)
            for msg, upd in zip(e.mpnn.fused_message_update, e.mpnn.update_fns):
                m = F.layer_norm(F.gelu(F.linear(torch.cat([x[row], x[col], attr], -1), msg[0].weight, msg[0].bias)), (c,))
                agg = torch.zeros_like(x).index_add_(0, col, m) / cnt[:, None]
                x = x + F.layer_norm(F.linear(torch.cat([x, agg], -1), upd[0].weight, upd[0].bias), (c,))
    x, p = x[tree_idx], pos[tree_idx]
    stack = []
    for i, L in enumerate(erwin.encoder):
        x = _layer_autograd(x, p, rots[i], L, W)
        stack.append((x, p, rots[i]))
        st = L.pool.stride
        if st != 1:
            with torch.no_grad():
                ctr = p.view(-1, st, d).mean(1)
                rel = (p.view(-1, st, d) - ctr[:, None]).reshape(-1, st * d)
            x = L.pool.norm(F.linear(torch.cat([x.reshape(-1, st * x.shape[1]), rel], 1), L.pool.proj.weight, L.pool.proj.bias))
            p = ctr
    x = _layer_autograd(x, p, rots[len(erwin.encoder)], erwin.bottleneck, W)
    if not erwin.decode:
        b = batch_idx[tree_idx]
        for s in erwin.strides:
            b = b[::s]
        return x, b
    for k, L in enumerate(erwin.decoder):
        cx, cp, crot = stack[len(erwin.encoder) - 1 - k]
        st = L.unpool.stride
        with torch.no_grad():
            rel = (cp.view(-1, st, d) - p[:, None]).reshape(-1, st * d)
        y = F.linear(torch.cat([x, rel], -1), L.unpool.proj.weight, L.unpool.proj.bias).reshape(cx.shape[0], -1)
        x = L.unpool.norm(cx + y)
        p = cp
        x = _layer_autograd(x, p, crot, L, W)
    mask = tree_mask.bool()
    return x[mask][torch.argsort(tree_idx[mask])]


def forward(erwin, feat, pos, batch_idx, edge_index=None, radius=None):
    needs_grad = torch.is_grad_enabled() and (feat.requires_grad or any(p.requires_grad for p in erwin.parameters()))
    if needs_grad:
        return forward_autograd(erwin, feat, pos, batch_idx, edge_index, radius)
    return forward_hip(erwin, feat, pos, batch_idx, edge_index, radius)
