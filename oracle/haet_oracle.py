"""TEST INFRASTRUCTURE ONLY -- CPU oracle for HAET's PhysicsAttention hot path.

A functional, fp32, PyTorch-CPU restatement of the reference's forward for the
one path this repo accelerates (Rep-Slice soft clustering + eidetic-state
reduction -> Erwin ball-tree attention over the slice tokens -> de-slice).  It
works on a plain ``state_dict`` (same keys as the reference modules) so that it
shares no code with the product modules in ``haet_amd/``.

Who may use it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- as the checker / the timed CPU baseline, never as a
fallback of the product path.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``tests/golden/make_golden.py`` and
committed as ``tests/golden/*.npz`` (checked by ``tests/test_oracle_golden.py``).
The ball-tree part is additionally pinned against the compiled reference header
(``oracle/_ref/libballtree_ref.so``).

Reference locations restated here (paths relative to the reference root):
  PA  = models/PhysicsAttention/{IrregularMesh,StructuredMesh2D,StructuredMesh3D}.py
  EF  = models/components/erwinflash/
  BT  = models/components/balltree/balltree.pyx
The attention inside a ball is *exact* softmax attention: the reference calls the
third-party ``flash_attn.flash_attn_qkvpacked_func`` (Dao-AILab/flash-attention,
not vendored, no version pinned by the reference; call site
EF/components/attention.py:82-84), whose published contract is
softmax(Q K^T * softmax_scale) V per (ball, head), no mask, no dropout.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import balltree as _bt

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class ErwinCfg:
    """Hyper-parameters of the Erwin stage (EF/erwin_flash.py:41-57)."""

    c_in: int
    c_hidden: List[int]
    ball_sizes: List[int]
    enc_num_heads: List[int]
    enc_depths: List[int]
    dec_num_heads: List[int]
    dec_depths: List[int]
    strides: List[int]
    rotate: float
    decode: bool = True
    mlp_ratio: int = 4
    dimensionality: int = 3


@dataclass
class AttnCfg:
    """Hyper-parameters of one Physics_Attention_* module (PA/IrregularMesh.py:41-63)."""

    dim: int
    heads: int = 8
    dim_head: int = 64
    slice_num: int = 64
    base_temp: float = 0.5
    epsilon: float = 1e-6
    kind: str = "irregular"  # irregular | structured2d | structured3d
    grid: tuple = ()  # (H, W) or (H, W, D) for the structured kinds
    kernel: int = 3
    erwin: Optional[ErwinCfg] = None
    rotate: float = 1
    mlp_ratio: int = 4

    def __post_init__(self):
        if self.erwin is None:
            self.erwin = default_erwin_cfg(self)


def default_erwin_cfg(a: AttnCfg) -> ErwinCfg:
    """Defaults used when the Erwin kwargs are None (PA/IrregularMesh.py:105-118);
    dimensionality is 3 for irregular/3-D and 2 for the 2-D module
    (PA/IrregularMesh.py:91, StructuredMesh2D.py:102, StructuredMesh3D.py:106)."""
    h, c, g = a.heads, a.dim_head, a.slice_num
    return ErwinCfg(
        c_in=c,
        c_hidden=[c, 2 * c],
        ball_sizes=[min(32, g), min(16, g // 2)],
        enc_num_heads=[h // 2, h],
        enc_depths=[2, 2],
        dec_num_heads=[h // 2],
        dec_depths=[2],
        strides=[2],
        rotate=a.rotate,
        decode=True,
        mlp_ratio=a.mlp_ratio,
        dimensionality=2 if a.kind == "structured2d" else 3,
    )


# --------------------------------------------------------------------------- ball tree
def rotation_matrix(angle_deg: float, dim: int) -> Tensor:
    """BT:576-596.  fp32 (torch.tensor of python floats); in 3-D one angle is used
    for all three Euler axes."""
    a = math.radians(angle_deg)
    c, s = math.cos(a), math.sin(a)
    if dim == 2:
        return torch.tensor([[c, -s], [s, c]])
    if dim == 3:
        return torch.tensor(
            [
                [c * c, s * c * (s - 1), s * (s + c * c)],
                [s * c, s * s * s + c * c, s * c * (s - 1)],
                [-s, s * c, c * c],
            ]
        )
    raise ValueError(f"Unsupported dimension: {dim}")


def build_balltree_with_rotations(pos: Tensor, batch_idx: Tensor, strides, ball_sizes, angle: float):
    """BT:598-662.  Returns (tree_idx int64[T'], tree_mask bool[T'], [rot_i | None])."""
    assert len(strides) == len(ball_sizes) - 1
    dim = pos.shape[1]
    bidx_np = batch_idx.detach().cpu().long().numpy()
    ti, tm = _bt.build_balltree(pos.detach().cpu().double().numpy(), bidx_np)
    tree_idx, tree_mask = torch.from_numpy(ti), torch.from_numpy(tm)
    n_levels = len(ball_sizes)
    if angle <= 0:
        return tree_idx, tree_mask, [None] * n_levels
    leaves = pos[tree_idx]
    cur_bidx = batch_idx[tree_idx]
    # NB: the TOTAL leaf count over all clouds enters here (reference quirk, BT:643)
    targets = [max(0, int(math.log2(leaves.shape[0] / bs))) for bs in ball_sizes]
    rotated = torch.matmul(leaves, rotation_matrix(angle, dim).to(leaves.dtype))
    rots = []
    for i in range(n_levels):
        r = _bt.partition_balltree(
            rotated.detach().cpu().double().numpy(), cur_bidx.cpu().long().numpy(), targets[i]
        )
        rots.append(torch.from_numpy(r))
        if i < n_levels - 1:
            rotated = rotated.view(-1, strides[i], dim).mean(dim=1)
            cur_bidx = cur_bidx[:: strides[i]]
    return tree_idx, tree_mask, rots


# --------------------------------------------------------------------------- Erwin pieces
def _lin(x: Tensor, sd: Dict[str, Tensor], name: str) -> Tensor:
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def rms_norm(x: Tensor, weight: Tensor) -> Tensor:
    """nn.RMSNorm(dim) with eps=None -> eps = finfo(dtype).eps (EF/components/layers.py:25-26)."""
    eps = torch.finfo(x.dtype).eps
    return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps) * weight


def batch_norm(x: Tensor, sd, name: str, training: bool) -> Tensor:
    """nn.BatchNorm1d (eps 1e-5): running stats in eval, biased batch stats over ALL
    tokens of all (b,h) clouds in train (EF/components/pooling.py:23,38,56,68)."""
    if training:
        mean, var = x.mean(dim=0), x.var(dim=0, unbiased=False)
    else:
        mean, var = sd[name + ".running_mean"], sd[name + ".running_var"]
    return (x - mean) * torch.rsqrt(var + 1e-5) * sd[name + ".weight"] + sd[name + ".bias"]


def ball_rel_pos(pos: Tensor, ball: int) -> Tensor:
    """EF/components/attention.py:35-40."""
    p = pos.view(-1, ball, pos.shape[1])
    return (p - p.mean(dim=1, keepdim=True)).reshape(-1, pos.shape[1])


def ball_msa(x: Tensor, pos: Tensor, sd, pre: str, heads: int, ball: int) -> Tensor:
    """EF/components/attention.py:42-91 in fp32 (no distance bias: sigma_att unused)."""
    x = x + _lin(ball_rel_pos(pos, ball), sd, pre + "pe_proj")
    c = x.shape[1]
    hd = c // heads
    qkv = _lin(x, sd, pre + "qkv").view(-1, ball, 3, heads, hd)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))  # [nb, h, n, hd]
    att = torch.softmax(torch.matmul(q, k.transpose(-1, -2)) * (1.0 / math.sqrt(hd)), dim=-1)
    o = torch.matmul(att, v).permute(0, 2, 1, 3).reshape(-1, c)
    return _lin(o, sd, pre + "proj")


def swiglu(x: Tensor, sd, pre: str) -> Tensor:
    """EF/components/mlp.py:8-18."""
    return _lin(_lin(x, sd, pre + "w2") * F.silu(_lin(x, sd, pre + "w1")), sd, pre + "w3")


def erwin_block(x: Tensor, pos: Tensor, sd, pre: str, heads: int, ball: int) -> Tensor:
    """EF/components/layers.py:30-32."""
    x = x + ball_msa(rms_norm(x, sd[pre + "norm1.weight"]), pos, sd, pre + "BMSA.", heads, ball)
    return x + swiglu(rms_norm(x, sd[pre + "norm2.weight"]), sd, pre + "swiglu.")


def run_blocks(x, pos, rot, sd, pre, depth, heads, ball, rotate_on, trace, tag):
    """EF/components/layers.py:72-92 (rotation on odd blocks, un-permute by argsort)."""
    use_rot = rotate_on and depth > 1
    inv = torch.argsort(rot) if use_rot else None
    for j in range(depth):
        bpre = f"{pre}blocks.{j}."
        if use_rot and j % 2 == 1:
            x = erwin_block(x[rot], pos[rot], sd, bpre, heads, ball)[inv]
        else:
            x = erwin_block(x, pos, sd, bpre, heads, ball)
        if trace is not None:
            trace[f"{tag}.block{j}"] = x
    return x


def ball_pool(x, pos, sd, pre, stride, training):
    """EF/components/pooling.py:25-40.  stride == 1: the node passes through unchanged
    (no projection, no BatchNorm: pooling.py:26-27)."""
    if stride == 1:
        return x, pos
    d = pos.shape[1]
    centers = pos.view(-1, stride, d).mean(dim=1)
    rel = (pos.view(-1, stride, d) - centers[:, None]).reshape(-1, stride * d)
    y = torch.cat([x.reshape(-1, stride * x.shape[1]), rel], dim=1)
    return batch_norm(_lin(y, sd, pre + "proj"), sd, pre + "norm", training), centers


def ball_unpool(x, pos, child_x, child_pos, sd, pre, stride, training):
    """EF/components/pooling.py:58-72."""
    d = pos.shape[1]
    rel = (child_pos.view(-1, stride, d) - pos[:, None]).reshape(-1, stride * d)
    y = _lin(torch.cat([x, rel], dim=-1), sd, pre + "proj").reshape(child_x.shape[0], -1)
    return batch_norm(child_x + y, sd, pre + "norm", training)


def erwin_forward(feat: Tensor, pos: Tensor, batch_idx: Tensor, sd, cfg: ErwinCfg,
                  pre: str = "erwin.", training: bool = False, trace: Optional[dict] = None,
                  tree=None) -> Tensor:
    """EF/erwin_flash.py:138-195 with embed=False, mp_steps=0 (the only configuration
    any PhysicsAttention module uses: PA/IrregularMesh.py:61-62,135-136)."""
    if tree is None:
        tree = build_balltree_with_rotations(pos, batch_idx, cfg.strides, cfg.ball_sizes, cfg.rotate)
    tree_idx, tree_mask, rots = tree
    if trace is not None:
        trace["tree_idx"], trace["tree_mask"] = tree_idx, tree_mask
        for i, r in enumerate(rots):
            if r is not None:
                trace[f"rot{i}"] = r
    rot_on = cfg.rotate > 0
    n_enc = len(cfg.enc_depths) - 1
    x, p = feat[tree_idx], pos[tree_idx]
    stack = []
    for i in range(n_enc):
        x = run_blocks(x, p, rots[i], sd, f"{pre}encoder.{i}.", cfg.enc_depths[i],
                       cfg.enc_num_heads[i], cfg.ball_sizes[i], rot_on, trace, f"enc{i}")
        stack.append((x, p, rots[i]))
        x, p = ball_pool(x, p, sd, f"{pre}encoder.{i}.pool.", cfg.strides[i], training)
        if trace is not None:
            trace[f"enc{i}.pool"] = x
    x = run_blocks(x, p, rots[n_enc], sd, f"{pre}bottleneck.", cfg.enc_depths[-1],
                   cfg.enc_num_heads[-1], cfg.ball_sizes[-1], rot_on, trace, "bott")
    if not cfg.decode:
        return x
    for k, i in enumerate(range(n_enc - 1, -1, -1)):
        cx, cp, crot = stack[i]
        x = ball_unpool(x, p, cx, cp, sd, f"{pre}decoder.{k}.unpool.", cfg.strides[i], training)
        p = cp
        if trace is not None:
            trace[f"dec{k}.unpool"] = x
        x = run_blocks(x, p, crot, sd, f"{pre}decoder.{k}.", cfg.dec_depths[i],
                       cfg.dec_num_heads[i], cfg.ball_sizes[i], rot_on, trace, f"dec{k}")
    return x[tree_mask][torch.argsort(tree_idx[tree_mask])]


# --------------------------------------------------------------------------- stand-alone ErwinFlash
def radius_graph(pos: Tensor, r: float, batch_idx: Tensor, max_num_neighbors: int = 32) -> Tensor:
    """torch_cluster.radius_graph(pos, r, batch=batch_idx, loop=True) as EF/erwin_flash.py:160-166 calls it
    (third-party, not vendored, no version pinned).  Published behaviour of its brute-force device kernel: for every
    centre i in index order, the first `max_num_neighbors` points j (index order, same cloud, i itself included)
    with |x_j - x_i|^2 < r^2 in fp32; flow source_to_target -> edge_index = [[j ...], [i ...]]."""
    n = pos.shape[0]
    rows, cols = [], []
    r2 = torch.tensor(r, dtype=torch.float32) ** 2
    for s in range(0, n, 2048):  # blocks of centres: n x n booleans would not fit for big clouds
        blk = pos[s:s + 2048]
        d2 = torch.zeros(blk.shape[0], n)
        for k in range(pos.shape[1]):  # the kernel's own accumulation order: dim 0, 1, 2
            diff = pos[None, :, k] - blk[:, None, k]
            d2 = d2 + diff * diff
        ok = (d2 < r2) & (batch_idx[s:s + 2048, None] == batch_idx[None, :])
        ok &= ok.long().cumsum(dim=1) <= max_num_neighbors
        i, j = ok.nonzero(as_tuple=True)
        rows.append(j)
        cols.append(i + s)
    return torch.stack([torch.cat(rows), torch.cat(cols)], dim=0)


def scatter_mean(src: Tensor, idx: Tensor, n: int) -> Tensor:
    """torch_scatter.scatter_mean(src, idx, dim=0, dim_size=n) (EF/components/utils/scatter_ops.py:5-18): mean of
    the rows per receiver, zero rows for receivers nobody points at."""
    acc = torch.zeros(n, src.shape[1], dtype=src.dtype).index_add_(0, idx, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, idx, torch.ones(idx.shape[0], dtype=src.dtype))
    return acc / cnt.clamp(min=1)[:, None]


def mpnn_forward(h: Tensor, pos: Tensor, edge_index: Tensor, sd, pre: str, steps: int) -> Tensor:
    """EF/components/mpnn.py:48-75: m_ij = LN(GELU(Linear([h_j', h_i', pos_j' - pos_i']))) with (j', i') =
    edge_index rows (0, 1), mean over the edges that END in a node, h += LN(Linear([h, m]))."""
    row, col = edge_index[0].long(), edge_index[1].long()
    edge_attr = pos[row] - pos[col]
    c = h.shape[1]
    for t in range(steps):
        m = _lin(torch.cat([h[row], h[col], edge_attr], dim=-1), sd, f"{pre}fused_message_update.{t}.0")
        m = F.layer_norm(F.gelu(m), (c,))
        agg = scatter_mean(m, col, h.shape[0])
        u = F.layer_norm(_lin(torch.cat([h, agg], dim=-1), sd, f"{pre}update_fns.{t}.0"), (c,))
        h = h + u
    return h


def erwinflash_forward(feat: Tensor, pos: Tensor, batch_idx: Tensor, sd, cfg: ErwinCfg, mp_steps: int,
                       edge_index: Optional[Tensor] = None, radius: Optional[float] = None,
                       embed: bool = True, training: bool = False, trace: Optional[dict] = None, tree=None) -> Tensor:
    """ErwinTransformer.forward on a raw point cloud (EF/erwin_flash.py:138-195) with the embedding
    (EF/components/embedding.py:9-20: Linear, then MPNN when mp_steps > 0)."""
    if embed:
        if edge_index is None and mp_steps > 0:
            assert radius is not None
            edge_index = radius_graph(pos, radius, batch_idx)
        x = _lin(feat, sd, "embed.embed_fn")
        if mp_steps > 0:
            x = mpnn_forward(x, pos, edge_index, sd, "embed.mpnn.", mp_steps)
    else:
        x = feat
    if trace is not None:
        trace["embed"], trace["edge_index"] = x, edge_index
    return erwin_forward(x, pos, batch_idx, sd, cfg, pre="", training=training, trace=trace, tree=tree)


# --------------------------------------------------------------------------- slice / de-slice
def project_points(x: Tensor, sd, cfg: AttnCfg) -> Tensor:
    """Per-point projection to [B, N, H*C] (PA/IrregularMesh.py:157-158,
    StructuredMesh2D.py:173-187, StructuredMesh3D.py:177-191)."""
    B, N, D = x.shape
    w, b = sd["in_project_x.weight"], sd.get("in_project_x.bias")
    if cfg.kind == "irregular":
        return F.linear(x, w, b)
    pad = cfg.kernel // 2
    if cfg.kind == "structured2d":
        H, W = cfg.grid
        g = x.reshape(B, H, W, D).permute(0, 3, 1, 2).contiguous()
        return F.conv2d(g, w, b, 1, pad).permute(0, 2, 3, 1).reshape(B, N, -1)
    H, W, Dd = cfg.grid
    g = x.reshape(B, H, W, Dd, D).permute(0, 4, 1, 2, 3).contiguous()
    return F.conv3d(g, w, b, 1, pad).permute(0, 2, 3, 4, 1).reshape(B, N, -1)


def slice_logits(x_proj: Tensor, sd, cfg: AttnCfg):
    """Ada-Temp + Rep-Slice logits (PA/IrregularMesh.py:168-173).  Returns
    (tau [B,H,N,1], logits/tau [B,H,N,G])."""
    tau = cfg.base_temp + _lin(x_proj, sd, "ada_temp_linear").clamp(min=-0.4, max=0.4)
    c0 = torch.log(-torch.log(torch.tensor(cfg.epsilon)))
    logits = _lin(x_proj, sd, "in_project_slice") - c0
    return tau, logits / tau


def token_positions(flat: Tensor, dimensionality: int) -> Tensor:
    """PA/IrregularMesh.py:204-209: min-max normalise the first `dimensionality`
    feature channels over ALL B*H*G tokens."""
    fmin = flat.min(dim=0, keepdim=True)[0]
    fmax = flat.max(dim=0, keepdim=True)[0]
    rng = fmax - fmin + 1e-8
    return (flat[:, :dimensionality] - fmin[:, :dimensionality]) / rng[:, :dimensionality]


def physics_attention_forward(x: Tensor, sd: Dict[str, Tensor], cfg: AttnCfg,
                              training: bool = False, trace: Optional[dict] = None) -> Tensor:
    """Forward of Physics_Attention_{Irregular_Mesh,Structured_Mesh_2D,_3D}
    (PA/IrregularMesh.py:141-235), fp32, dropout p=0."""
    B, N, _ = x.shape
    H, C, G = cfg.heads, cfg.dim_head, cfg.slice_num
    x_lin = project_points(x, sd, cfg)
    x_proj = x_lin.reshape(B, N, H, C).permute(0, 2, 1, 3).contiguous()
    tau, z = slice_logits(x_proj, sd, cfg)
    w = torch.softmax(z, dim=2)  # over the POINT axis (dim=2 of [B,H,N,G])
    norm = w.sum(2)
    S = torch.einsum("bhnc,bhng->bhgc", x_proj, w)
    S = S / (norm + 1e-5)[:, :, :, None].repeat(1, 1, 1, C)
    flat = S.reshape(B * H * G, C)
    # Positions carry no gradient: every use of them in the reference is under no_grad
    # (attention.py:35 compute_rel_pos is @torch.no_grad; pooling.py:29,59 centres / rel_pos under
    # torch.no_grad(); the tree build is no_grad too, erwin_flash.py:149).
    pos = token_positions(flat, cfg.erwin.dimensionality).detach()
    batch_idx = torch.arange(B * H).repeat_interleave(G)
    etrace = {} if trace is not None else None
    flat2 = erwin_forward(flat, pos, batch_idx, sd, cfg.erwin, "erwin.", training, etrace)
    S2 = flat2.reshape(B, H, G, C)
    out_x = torch.einsum("bhgc,bhng->bhnc", S2, w)
    out_x = out_x.permute(0, 2, 1, 3).reshape(B, N, H * C)
    out = _lin(out_x, sd, "to_out.0")
    if trace is not None:
        m = z.max(dim=2)[0]
        trace.update(
            x_lin=x_lin, tau=tau.squeeze(-1), col_max=m,
            col_sumexp=torch.exp(z - m[:, :, None, :]).sum(2),
            eidetic=S, pos=pos, batch_idx=batch_idx, processed=S2, out_x=out_x, out=out,
        )
        trace.update({"erwin." + k: v for k, v in etrace.items()})
    return out


# --------------------------------------------------------------------------- callers (API surface)
def layer_norm(x, sd, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def mlp0(x, sd, pre):
    """models/components/mlp.py MLP with n_layers=0, gelu (the only form the blocks use)."""
    return _lin(F.gelu(_lin(x, sd, pre + "linear_pre.0")), sd, pre + "linear_post")


def block_forward(fx, sd, pre, cfg: AttnCfg, last: bool, training=False):
    """TransolverErwinBlock.forward (models/HAETransolver_Irregular_Mesh.py:114-134)."""
    asd = {k[len(pre) + 5:]: v for k, v in sd.items() if k.startswith(pre + "Attn.")}
    fx = physics_attention_forward(layer_norm(fx, sd, pre + "ln_1"), asd, cfg, training) + fx
    fx = mlp0(layer_norm(fx, sd, pre + "ln_2"), sd, pre + "mlp.") + fx
    if last:
        return _lin(layer_norm(fx, sd, pre + "ln_3"), sd, pre + "mlp2")
    return fx


def model_forward(x, fx, sd, cfg: AttnCfg, n_layers: int, training=False):
    """Model.forward (models/HAETransolver_Irregular_Mesh.py:338-376), unified_pos=False,
    T=None; for the structured models the placeholder is only added when fx is None
    (models/HAETransolver_Structured_Mesh_2D.py:399-402)."""
    if fx is not None:
        h = mlp0(torch.cat((x, fx), -1), sd, "preprocess.")
        if cfg.kind == "irregular":
            h = h + sd["placeholder"][None, None, :]
    else:
        h = mlp0(x, sd, "preprocess.") + sd["placeholder"][None, None, :]
    for k in range(n_layers):
        h = block_forward(h, sd, f"blocks.{k}.", cfg, k == n_layers - 1, training)
    return h


def rel_l2(a: Tensor, b: Tensor) -> float:
    """||a-b||_2 / ||b||_2 (the metric of BASELINE.json / utils/testloss.py:31-45)."""
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
