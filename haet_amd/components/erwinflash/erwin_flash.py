"""Erwin hierarchical ball transformer over the slice tokens -- host-side mirror of
models/components/erwinflash/ (ErwinTransformer erwin_flash.py:14-195, BasicLayer /
ErwinTransformerBlock components/layers.py, BallMSA components/attention.py, SwiGLU
components/mlp.py, BallPooling / BallUnpooling components/pooling.py).

The module tree exists to own the parameters under the reference's exact state_dict
keys; the arithmetic of the whole stage is ONE fused HIP kernel (haet_erwin_fwd) fed by
the device ball-tree builder (haet_balltree_build), instead of ~100 small launches and
a host round trip per forward.  That is the configuration every PhysicsAttention module
uses (embed=False, mp_steps=0, decode=True, <= a few hundred tokens per cloud); raw point clouds (embedding, MPNN,
clouds of any size) take the streaming path of pointcloud.py.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from ... import _lib, ops
from ..._lib import HaetError


class SwiGLU(nn.Module):
    """W3(W2 x * silu(W1 x))  (components/mlp.py:8-18)."""

    def __init__(self, in_dim: int, dim: int):
        super().__init__()
        self.w1 = nn.Linear(in_dim, dim)
        self.w2 = nn.Linear(in_dim, dim)
        self.w3 = nn.Linear(dim, in_dim)


class BallMSA(nn.Module):
    """Ball attention parameters (components/attention.py:13-25). sigma_att is kept for
    checkpoint compatibility; like the reference's flash variant it is never used."""

    def __init__(self, dim: int, num_heads: int, ball_size: int, dimensionality: int = 3):
        super().__init__()
        self.num_heads, self.ball_size = num_heads, ball_size
        self.head_dim = dim // num_heads
        assert self.head_dim * num_heads == dim, "dim must be divisible by num_heads"
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)
        self.pe_proj = nn.Linear(dimensionality, dim)
        self.sigma_att = nn.Parameter(-1 + 0.01 * torch.randn((1, num_heads, 1, 1)))


class ErwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, ball_size, mlp_ratio, dimensionality=3):
        super().__init__()
        self.ball_size = ball_size
        self.norm1 = nn.RMSNorm(dim)
        self.norm2 = nn.RMSNorm(dim)
        self.BMSA = BallMSA(dim, num_heads, ball_size, dimensionality)
        self.swiglu = SwiGLU(dim, dim * mlp_ratio)


class BallPooling(nn.Module):
    def __init__(self, in_dim, out_dim, stride, dimensionality=3):
        super().__init__()
        self.stride = stride
        self.proj = nn.Linear(stride * in_dim + stride * dimensionality, out_dim)
        self.norm = nn.BatchNorm1d(out_dim)


class BallUnpooling(nn.Module):
    def __init__(self, in_dim, out_dim, stride, dimensionality=3):
        super().__init__()
        self.stride = stride
        self.proj = nn.Linear(in_dim + stride * dimensionality, stride * out_dim)
        self.norm = nn.BatchNorm1d(out_dim)


class BasicLayer(nn.Module):
    """components/layers.py:35-70 (parameters only)."""

    def __init__(self, direction, depth, stride, in_dim, out_dim, num_heads, ball_size,
                 mlp_ratio, rotate, dimensionality=3):
        super().__init__()
        hidden_dim = in_dim if direction == "down" else out_dim
        self.hidden_dim, self.num_heads, self.ball_size = hidden_dim, num_heads, ball_size
        self.mlp_ratio = mlp_ratio
        self.blocks = nn.ModuleList(
            [ErwinTransformerBlock(hidden_dim, num_heads, ball_size, mlp_ratio, dimensionality)
             for _ in range(depth)])
        self.rotate = [i % 2 for i in range(depth)] if rotate else [False] * depth
        if direction == "down" and stride is not None:
            self.pool = BallPooling(hidden_dim, out_dim, stride, dimensionality)
        elif direction == "up" and stride is not None:
            self.unpool = BallUnpooling(in_dim, hidden_dim, stride, dimensionality)


def _equal_clouds(batch_idx: torch.Tensor, n_clouds: int) -> bool:
    M = batch_idx.shape[0] // n_clouds
    return bool((batch_idx.view(n_clouds, M) == torch.arange(n_clouds, device=batch_idx.device)[:, None]).all())


class MPNN(nn.Module):
    """Parameters of the message-passing embedding (components/mpnn.py:21-46): per step a message Linear(2 dim + d -> dim)
    [+ GELU + LayerNorm without affine] and an update Linear(2 dim -> dim) [+ LayerNorm without affine]."""

    def __init__(self, dim: int, mp_steps: int, dimensionality: int = 3):
        super().__init__()
        self.fused_message_update = nn.ModuleList(
            [nn.Sequential(nn.Linear(2 * dim + dimensionality, dim), nn.GELU(), nn.LayerNorm(dim, elementwise_affine=False))
             for _ in range(mp_steps)])
        self.update_fns = nn.ModuleList(
            [nn.Sequential(nn.Linear(2 * dim, dim), nn.LayerNorm(dim, elementwise_affine=False)) for _ in range(mp_steps)])


class ErwinEmbedding(nn.Module):
    """Linear projection -> MPNN (components/embedding.py:9-20), parameters only."""

    def __init__(self, in_dim: int, dim: int, mp_steps: int, dimensionality: int = 3):
        super().__init__()
        self.mp_steps = mp_steps
        self.embed_fn = nn.Linear(in_dim, dim)
        self.mpnn = MPNN(dim, mp_steps, dimensionality)


def _mark_stale_hook(module, incompatible):
    module._mark_stale()


def _barrier_check_hook(module, prefix, keep_vars):
    # state_dict() copies the running statistics out: a natural point to learn that a train-mode launch's grid barrier
    # gave up (the statistics would be garbage) -- raises HaetError
    module.check_grid_barrier()


class ErwinTransformer(nn.Module):
    """Same constructor and state_dict as the reference (erwin_flash.py:41-136)."""

    def __init__(self, c_in: int, c_hidden: List, ball_sizes: List, enc_num_heads: List,
                 enc_depths: List, dec_num_heads: List, dec_depths: List, strides: List,
                 rotate: int, decode: bool = True, mlp_ratio: int = 4, dimensionality: int = 3,
                 mp_steps: int = 3, embed: bool = True):
        super().__init__()
        assert len(enc_num_heads) == len(enc_depths) == len(ball_sizes)
        assert len(dec_num_heads) == len(dec_depths) == len(strides)
        assert len(strides) == len(ball_sizes) - 1
        self.rotate, self.decode = rotate, decode
        self.ball_sizes, self.strides = list(ball_sizes), list(strides)
        self.mp_steps, self.use_embedding = mp_steps, embed
        self.dimensionality = dimensionality
        self.c_hidden = list(c_hidden)
        if embed:  # raw point clouds (stand-alone use, erwin_flash.py:69-71): Linear + MPNN in front of the tree
            self.embed = ErwinEmbedding(c_in, c_hidden[0], mp_steps, dimensionality)
        num_layers = len(enc_depths) - 1
        self.encoder = nn.ModuleList()
        for i in range(num_layers):
            self.encoder.append(BasicLayer("down", enc_depths[i], strides[i], c_hidden[i], c_hidden[i + 1],
                                           enc_num_heads[i], ball_sizes[i], mlp_ratio, rotate > 0, dimensionality))
        self.bottleneck = BasicLayer(None, enc_depths[-1], None, c_hidden[-1], c_hidden[-1],
                                     enc_num_heads[-1], ball_sizes[-1], mlp_ratio, rotate > 0, dimensionality)
        if decode:
            self.decoder = nn.ModuleList()
            for i in range(num_layers - 1, -1, -1):
                self.decoder.append(BasicLayer("up", dec_depths[i], strides[i], c_hidden[i + 1], c_hidden[i],
                                               dec_num_heads[i], ball_sizes[i], mlp_ratio, rotate > 0, dimensionality))
        self.in_dim, self.out_dim = c_in, c_hidden[0]
        self.apply(self._init_weights)
        self._packed = None  # (key, weights blob, desc)
        self.register_load_state_dict_post_hook(_mark_stale_hook)  # (module-level functions: the module stays picklable)
        self.register_state_dict_pre_hook(_barrier_check_hook)

    def _init_weights(self, m):
        # erwin_flash.py:127-136
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, mean=0.0, std=0.02, a=-2.0, b=2.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
            if m.weight is not None:
                nn.init.constant_(m.weight, 1.0)

    # ------------------------------------------------------------------ packing
    def _pack_key(self):
        # Walking nn.Module.parameters()/buffers() costs ~0.1 ms per forward, so the tensor list is
        # cached.  The tensor OBJECTS only change through .to()/.cuda() (-> _apply), load_state_dict
        # with assign=True (-> post hook, which also fires for an ancestor's load) or a parameter / buffer
        # re-assigned by hand on a submodule (-> torch's global registration hooks bump
        # _lib.cache_generation(), seen here at the very next forward).
        ts = self._tensor_list()
        # (_stats_epoch: train-mode launches update the BatchNorm running statistics through raw pointers, which bumps no
        # `_version` -- an eval-mode blob folded from the old statistics must not survive them)
        return tuple([(t.data_ptr(), t._version) for t in ts] + [(-1, _lib.invalidate_epoch()), (-2, self.__dict__.get("_stats_epoch", 0))])

    def _mark_stale(self, *_):
        self.__dict__["_pack_stale"] = True

    def __getstate__(self):
        # weight-derived caches hold ctypes descriptors and device scratch: never pickled, rebuilt at the next forward
        state = dict(self.__dict__)
        for k in ("_train_cache", "_packed_f32", "_packed_frags", "_fits_memo", "_pack_tensors", "_pack_params", "_plans_hit", "_flat_hit", "_stats_epoch", "last_positions"):
            state.pop(k, None)
        state["_packed"] = None
        state["_pack_stale"] = True
        return state

    def invalidate_caches(self):
        """Drop the packed weight blob (rebuilt at the next forward): needed after in-place writes through
        `.data` (EMA swap, manual clipping), which change neither `_version` nor the storage pointer."""
        self._packed = None
        self.__dict__.pop("_packed_frags", None)
        self._mark_stale()

    def _apply(self, fn, *args, **kwargs):
        self._mark_stale()
        return super()._apply(fn, *args, **kwargs)

    def packed(self):
        """(weights blob, desc) for haet_erwin_fwd, rebuilt only when a parameter changed."""
        key = (self._pack_key(), self.training)
        if self._packed is None or self._packed[0] != key:
            plan = self._pack_plan("fwd", self.training)
            self._packed = (key, plan.run(self._flat(key[0])), plan.desc)
        return self._packed[1], self._packed[2]

    def _tensor_list(self):
        """The cached list of parameters + buffers (see _pack_key); a NEW list object whenever the tensor objects may have
        changed -- what the plans below are keyed on."""
        d = self.__dict__
        ts = d.get("_pack_tensors")
        gen = _lib.cache_generation()
        if ts is None or d.get("_pack_stale", True) or d.get("_pack_gen") != gen:
            ps = list(self.parameters())
            ts = ps + list(self.buffers())
            d["_pack_tensors"], d["_pack_params"], d["_pack_stale"], d["_pack_gen"] = ts, ps, False, gen
        return ts

    def _plans(self):
        ts = self._tensor_list()
        hit = self.__dict__.get("_plans_hit")
        if hit is None or hit[0] is not ts:
            from .pack import _Sources

            hit = (ts, {}, _Sources())
            self.__dict__["_plans_hit"] = hit
        return hit

    def _pack_plan(self, which: str, mode):
        """The blobs' layouts as gather plans (pack.py ErwinPackPlan / ErwinF32PackPlan): built once per set of parameter
        OBJECTS (and mode) over ONE source registry, so that a repack after an optimizer step is one concatenation of the
        parameters and one gather per blob (~9 launches) instead of ~400 per blob."""
        from .pack import ErwinF32PackPlan, ErwinPackPlan

        _, plans, src = self._plans()
        if (which, mode) not in plans:  # (train and eval plans side by side: a validation pass per epoch rebuilds neither)
            plans[(which, mode)] = (ErwinPackPlan if which == "fwd" else ErwinF32PackPlan)(self, src)
        return plans[(which, mode)]

    def _flat(self, key):
        """The flat source vector of the plans for the weights `key` describes (shared by the two blobs of a training step)."""
        src = self._plans()[2]
        hit = self.__dict__.get("_flat_hit")
        if hit is None or hit[0] != key or hit[1] != src.n:
            hit = (key, src.n, src.flat(self._tensor_list()[0].device))
            self.__dict__["_flat_hit"] = hit
        return hit[2]

    def _stats_updated(self):
        self.__dict__["_stats_epoch"] = self.__dict__.get("_stats_epoch", 0) + 1

    def param_list(self):
        """list(self.parameters()) without the module-tree walk (the cached list of _tensor_list)."""
        self._tensor_list()
        return self.__dict__["_pack_params"]

    def plan_desc(self):
        """The descriptor alone (offsets and hyper-parameters: independent of the weights' values) -- no key walk, no repack."""
        return self._pack_plan("fwd", self.training).desc

    def packed_frags(self, M: int):
        """The fragment-order weight blob of the fragment kernel (csrc/erwin2.hip) for clouds of M tokens, or None
        outside its envelope; rebuilt (on the device) only when the packed blob was."""
        blob, desc = self.packed()
        hit = self.__dict__.get("_packed_frags")
        if hit is None or hit[0] is not blob or hit[1] != M:
            hit = (blob, M, ops.erwin_frags(blob, desc, M))
            self.__dict__["_packed_frags"] = hit
        return hit[2]

    # ------------------------------------------------------------------ forward
    def _token_kernel_fits(self, T: int, n_clouds: int) -> bool:
        """The one-launch LDS-resident kernel (csrc/erwin.hip): equally sized clouds whose whole hierarchy fits one CU."""
        if self.use_embedding or not self.decode or T % n_clouds:
            return False
        _, desc = self.packed()
        memo = self.__dict__.get("_fits_memo")
        if memo is None or memo[0] is not desc:  # (a new desc object = repacked weights)
            memo = (desc, {})
            self.__dict__["_fits_memo"] = memo
        hit = memo[1].get((T, n_clouds))
        if hit is None:
            need = _lib.load().haet_erwin_lds_bytes(desc, ops.tree_slots(T // n_clouds))
            hit = memo[1][(T, n_clouds)] = 0 < need <= 160 * 1024
        return hit

    def forward(self, node_features: torch.Tensor, node_positions: Optional[torch.Tensor],
                batch_idx: Optional[torch.Tensor] = None, n_clouds: Optional[int] = None,
                edge_index: Optional[torch.Tensor] = None, tree_idx=None, tree_mask=None, radius: Optional[float] = None,
                **kwargs):
        """Two regimes behind the reference's signature (erwin_flash.py:138-148):
        * slice tokens (what PhysicsAttention passes: `n_clouds` equally sized clouds of <= a few hundred tokens, no
          embedding): tokens [T, c], positions [T, d] or None (= min-max normalised first `dimensionality` channels of
          the tokens, IrregularMesh.py:204-209, computed inside the launch; kept in `last_positions`) -> ONE launch;
        * raw point clouds (embed / MPNN, clouds of any and unequal size, decode=False): node_features [N, c_in],
          node_positions [N, d], batch_idx [N] (+ edge_index or radius) -> the streaming path of pointcloud.py.
        A caller-made tree (tree_idx / tree_mask) is not accepted: the reference itself fails on it (its rotated
        partitions are only defined on the path that builds the tree, erwin_flash.py:151-157,184)."""
        if tree_idx is not None or tree_mask is not None:
            raise NotImplementedError("tree_idx / tree_mask from the caller: the tree is always built on the device")
        T = node_features.shape[0]
        if n_clouds is None:
            if batch_idx is None:
                raise HaetError("either batch_idx or n_clouds is required")
            if node_positions is None or self.use_embedding or not self.decode:
                n_clouds = -1
            else:
                n_clouds = int(batch_idx[-1].item()) + 1  # sync; PhysicsAttention passes n_clouds
        if n_clouds < 0 or not self._token_kernel_fits(T, n_clouds) or (
                batch_idx is not None and node_positions is not None and n_clouds > 1
                and kwargs.get("check_cloud_sizes", True) and not _equal_clouds(batch_idx, n_clouds)):
            from . import pointcloud

            if node_positions is None or batch_idx is None:
                raise HaetError("the point-cloud path needs node_positions and batch_idx")
            return pointcloud.forward(self, node_features, node_positions, batch_idx, edge_index, radius)
        M = T // n_clouds
        blob, desc = self.packed()
        train = None
        if self.training:  # BatchNorm batch statistics over all clouds + running-stat update, inside the kernel
            if n_clouds > _lib.load().haet_resident_workgroup_limit():
                # more clouds than compute units: the kernel's grid barrier could not complete -- the torch
                # restatement of the stage (same tree, same arithmetic, launch-bound) takes this case
                from ... import _autograd

                if node_positions is None:
                    node_positions = ops.token_pos(node_features, self.dimensionality)
                self.__dict__["last_positions"] = node_positions if self.__dict__.get("keep_positions", True) else None
                with torch.no_grad():
                    return _autograd.erwin_forward(self, node_features, node_positions, n_clouds)
            train = self._train_state(desc, n_clouds, node_features.device)
            if os.environ.get("HAET_CHECK_BARRIER"):  # debug: learn about a timed-out barrier at every launch (syncs)
                train.check_barrier()
        if node_positions is None and not os.environ.get("HAET_NO_TOKEN_STAGE") and ops.token_stage_supported(desc, n_clouds, M):
            out, node_positions = ops.token_stage(node_features, blob, desc, n_clouds, M, self.dimensionality,
                                                  self.strides, float(self.rotate), train,
                                                  frags=None if train is not None else self.packed_frags(M))
            if train is not None:
                self._stats_updated()
            self.__dict__["last_positions"] = node_positions if self.__dict__.get("keep_positions", True) else None
            return out
        if node_positions is None:
            tree_idx, tree_mask, rot, _, node_positions = ops.balltree_build(
                None, n_clouds, M, self.strides, self.ball_sizes, float(self.rotate),
                tokens=node_features, d=self.dimensionality)
        else:
            tree_idx, tree_mask, rot, _ = ops.balltree_build(
                node_positions, n_clouds, M, self.strides, self.ball_sizes, float(self.rotate))
        self.__dict__["last_positions"] = node_positions if self.__dict__.get("keep_positions", True) else None
        if train is not None:
            self._stats_updated()
        return ops.erwin_fwd(node_features, node_positions, tree_idx, tree_mask, rot, blob, desc, n_clouds, M, train)

    # ------------------------------------------------------------------ training: forward with a tape, HIP backward
    def packed_f32(self):
        """(fp32 blob, layout, [(parameter, offset)]) for haet_erwin_bwd, rebuilt only when a parameter changed."""
        key = self._pack_key()
        hit = self.__dict__.get("_packed_f32")
        if hit is None or hit[0] != key:
            plan = self._pack_plan("f32", None)
            hit = (key, plan.run(self._flat(key)), plan.lay, plan.pmap)
            self.__dict__["_packed_f32"] = hit
        return hit[1], hit[2], hit[3]

    def hip_backward_supported(self, n_clouds: int, M: int) -> bool:
        if os.environ.get("HAET_TORCH_ERWIN_BWD") or not self.decode:
            return False
        return ops.erwin_bwd_supported(self.plan_desc(), n_clouds, M)

    def forward_taped(self, tokens: torch.Tensor, n_clouds: int):
        """Forward of the stage (positions from the tokens, tree, Erwin; BatchNorm per self.training) that also records
        what haet_erwin_bwd needs.  Returns (tokens', ctx)."""
        T = tokens.shape[0]
        M = T // n_clouds
        blob, desc = self.packed()
        # the backward's fp32 blob is made NOW: the weights are the forward's (nothing steps between a forward and its
        # backward), the flat source vector is the one the forward blob was just gathered from, and the backward then
        # walks no parameter list at all
        f32 = self.packed_f32()
        st = self._train_state(desc, n_clouds, tokens.device)  # (eval: only its tape slot is used)
        tape, opf = ops.erwin_tape(desc, n_clouds, M, tokens.device)
        st.set_tape(tape, opf)
        try:
            if not os.environ.get("HAET_NO_TOKEN_STAGE") and ops.token_stage_supported(desc, n_clouds, M):
                out, pos, tree_idx, tree_mask, rot, _ = ops.token_stage(tokens, blob, desc, n_clouds, M, self.dimensionality,
                                                                        self.strides, float(self.rotate), st, want_tree=True)
            else:
                tree_idx, tree_mask, rot, _, pos = ops.balltree_build(None, n_clouds, M, self.strides, self.ball_sizes,
                                                                      float(self.rotate), tokens=tokens, d=self.dimensionality)
                out = ops.erwin_fwd(tokens, pos, tree_idx, tree_mask, rot, blob, desc, n_clouds, M, st)
        finally:
            st.set_tape(None)
        stats = st.batch_stats.clone() if self.training else None
        if self.training:
            self._stats_updated()
        ctx = dict(tape=tape, pos=pos, tree_idx=tree_idx, tree_mask=tree_mask, rot=rot, blob=blob, desc=desc,
                   n_clouds=n_clouds, M=M, stats=stats, workspace=st.workspace, f32=f32)
        return out, ctx

    def backward_hip(self, ctx, g_out: torch.Tensor):
        """haet_erwin_bwd on a forward_taped context: (dL/d tokens, {parameter: gradient})."""
        blob32, lay, pmap = ctx["f32"] if "f32" in ctx else self.packed_f32()

        class _St:  # what ops.erwin_bwd reads of a train state
            batch_stats, workspace = ctx["stats"], ctx["workspace"]

        g_in, per_cloud = ops.erwin_bwd(g_out, ctx["tape"], ctx["pos"], ctx["tree_idx"], ctx["tree_mask"], ctx["rot"], blob32,
                                        lay.total, ctx["blob"], ctx["desc"], ctx["n_clouds"], ctx["M"],
                                        _St if ctx["stats"] is not None else None)
        gsum = per_cloud.sum(dim=0)
        return g_in, {id(p): gsum[off:off + p.numel()].view_as(p) for p, off in pmap}

    def train_state(self):
        """The ops.ErwinTrainState of the last train-mode / taped launch, or None."""
        hit = self.__dict__.get("_train_cache")
        return None if hit is None else hit[1]

    def check_grid_barrier(self):
        """Raise HaetError if a grid barrier of a train-mode launch of this module timed out since the last check
        (ops.ErwinTrainState.check_barrier; synchronises).  Called from state_dict(), from the owning attention module
        whenever it re-reads its parameters after an optimizer step, and per launch under HAET_CHECK_BARRIER=1."""
        hit = self.__dict__.get("_train_cache")
        if hit is not None and hit[1].workspace.is_cuda:
            hit[1].check_barrier()

    def _train_state(self, desc, n_clouds, device):
        """ops.ErwinTrainState for this (weights object set, cloud count): pointers to the BatchNorm running
        statistics (updated in place by the kernel), batch-statistics output, grid-barrier workspace."""
        ts = self._tensor_list()
        key = (n_clouds, str(device), tuple(b.data_ptr() for b in ts[len(self.__dict__["_pack_params"]):]))  # (the buffers)
        hit = self.__dict__.get("_train_cache")
        if hit is None or hit[0] != key:
            hit = (key, ops.ErwinTrainState(self, desc, n_clouds, device))
            self.__dict__["_train_cache"] = hit
        return hit[1]
