"""Shared body of the three HAETransolver_* model wrappers.

Constructor kwargs, forward(x, fx, T) and state_dict keys follow
models/HAETransolver_Irregular_Mesh.py:26-376 (and the Structured_Mesh_2D / _3D twins).  The attention
inside every block is the HIP path; the LayerNorm / MLP layers around it are stock PyTorch-ROCm modules
that, with gradients disabled, also run on HIP kernels (haet_amd/_fast.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.utils.checkpoint as checkpoint
from torch.nn.init import trunc_normal_

from . import _compile, _fast
from .components import MLP, timestep_embedding


class TransolverErwinBlockBase(_compile.OpaqueForwardMixin, nn.Module):
    """LN -> Attn -> +res -> LN -> MLP -> +res [-> LN -> Linear]
    (HAETransolver_Irregular_Mesh.py:114-134)."""

    def _init_block(self, attn: nn.Module, hidden_dim, act, mlp_ratio, last_layer, out_dim):
        self.last_layer = last_layer
        self.ln_1 = nn.LayerNorm(hidden_dim)
        self.Attn = attn
        self.ln_2 = nn.LayerNorm(hidden_dim)
        self.mlp = MLP(hidden_dim, hidden_dim * mlp_ratio, hidden_dim, n_layers=0, res=False, act=act)
        if self.last_layer:
            self.ln_3 = nn.LayerNorm(hidden_dim)
            self.mlp2 = nn.Linear(hidden_dim, out_dim)
        _compile.register(self)

    def forward(self, fx):
        # torch.compile without gradients: the whole block is one opaque operator (haet_amd/_compile.py)
        if _compile.use_operator(fx):
            out = self.mlp2.out_features if self.last_layer else fx.shape[-1]
            return torch.ops.haet.module_forward(fx, _compile.primed_handle(self._haet_handle), out)
        return self._forward_eager(fx)

    def _opaque_forward(self, fx):
        return self._forward_eager(fx)

    def _prime(self):
        p = next(self.parameters())
        if p.is_cuda:
            was = self.training
            self.eval()
            try:
                self._forward_eager(torch.zeros(1, self.Attn._prime_points(), self.ln_1.normalized_shape[0], device=p.device))
            finally:
                self.train(was)

    @torch.compiler.disable
    def _forward_eager(self, fx):
        if _fast.usable(fx):  # LayerNorms / MLP on the HIP kernels too, both directions (haet_amd/_fast.py)
            if _fast.enabled(fx):
                # inference: ln_1 rides in the projection GEMM's loader, ln_2 in the MLP's first GEMM's
                # ... and the residual in the epilogue of the fused de-slice + to_out pass
                fx = self.Attn(fx, pre_norm=self.ln_1, residual=True)
                m = _fast.mlp(self.mlp, fx, pre_norm=self.ln_2)
                if m is None:
                    m = self.mlp(_fast.layer_norm(self.ln_2, fx))
                if self.last_layer:
                    return _fast.add_layer_norm_linear(self.ln_3, self.mlp2, m, fx)
                return m.add_(fx)
            fx = self.Attn(_fast.layer_norm(self.ln_1, fx)) + fx
            m = self.mlp(_fast.layer_norm(self.ln_2, fx))
            if self.last_layer:
                return _fast.linear(self.mlp2, _fast.layer_norm(self.ln_3, m + fx))
            return m + fx
        fx = self.Attn(self.ln_1(fx)) + fx
        fx = self.mlp(self.ln_2(fx)) + fx
        if self.last_layer:
            return self.mlp2(self.ln_3(fx))
        return fx


class _PreprocessOp:
    """Handle target of haet::module_forward for a model's preprocess step (not an nn.Module: it must not appear in
    the model's state_dict or module tree)."""

    def __init__(self, model, with_placeholder: bool):
        import weakref

        self._model = weakref.ref(model)
        self._with = with_placeholder

    def _opaque_forward(self, inp):
        return self._model()._preprocess(inp, self._with)

    def _prime(self):
        m = self._model()
        lin = m.preprocess.linear_pre[0]
        if lin.weight.is_cuda:
            m._preprocess(torch.zeros(1, 256, lin.in_features, device=lin.weight.device), self._with)


class ModelBase(_compile.OpaqueForwardMixin, nn.Module):
    placeholder_always = False  # irregular: always added; structured: only when fx is None

    def _init_model(self, make_block, in_dim, n_layers, n_hidden, act, Time_Input):
        self.preprocess = MLP(in_dim, n_hidden * 2, n_hidden, n_layers=0, res=False, act=act)
        if Time_Input:
            self.time_fc = nn.Sequential(nn.Linear(n_hidden, n_hidden), nn.SiLU(), nn.Linear(n_hidden, n_hidden))
        self.blocks = nn.ModuleList([make_block(i == n_layers - 1) for i in range(n_layers)])
        self.initialize_weights()
        self.placeholder = nn.Parameter((1 / n_hidden) * torch.rand(n_hidden, dtype=torch.float))
        self._after_copy()

    def _after_copy(self):
        # the preprocess step as opaque operators for torch.compile (without / with the placeholder)
        pre = (_PreprocessOp(self, False), _PreprocessOp(self, True))
        self.__dict__["_pre_ops"] = pre  # (strong references: the registry only holds weak ones)
        self.__dict__["_pre_handles"] = (_compile.register(pre[0]), _compile.register(pre[1]))

    def __getstate__(self):
        # the preprocess operators hold a weak reference to the model: re-made by _after_copy() on unpickling
        state = dict(self.__dict__)
        for k in ("_pre_ops", "_pre_handles", "_haet_handle"):
            state.pop(k, None)
        return state

    def initialize_weights(self):
        self.apply(self._init_weights)

    @torch.compiler.disable
    def _preprocess_eager(self, inp, with_placeholder: bool):
        return self._preprocess(inp, with_placeholder)

    def _init_weights(self, m):
        # HAETransolver_Irregular_Mesh.py:279-295
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm1d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def invalidate_caches(self):
        """Re-derive every cached weight operand of the model (attention modules and the point-wise fast path) at
        the next forward -- for in-place parameter writes through `.data` (see PhysicsAttentionBase.invalidate_caches)."""
        from . import _lib

        _lib.invalidate_all_caches()
        for m in self.modules():
            if m is not self and hasattr(m, "invalidate_caches"):
                m.invalidate_caches()

    def unified_positions(self, x):
        raise NotImplementedError

    def _preprocess(self, inp, with_placeholder: bool):
        """preprocess MLP (+ placeholder): HIP fast path where it applies (inference: the placeholder rides in the last
        Linear's bias), stock modules otherwise."""
        fast = _fast.mlp(self.preprocess, inp, self.placeholder if with_placeholder else None)
        if fast is not None:
            return fast
        fx = self.preprocess(inp)
        if with_placeholder:
            fx = fx + self.placeholder[None, None, :]
        return fx

    def forward(self, x, fx, T=None):
        if self.unified_pos:
            x = self.unified_positions(x)
        inp = torch.cat((x, fx), -1) if fx is not None else x
        with_placeholder = fx is None or self.placeholder_always
        if _compile.use_operator(inp):  # torch.compile without gradients: one opaque operator (haet_amd/_compile.py)
            fx = torch.ops.haet.module_forward(inp, _compile.primed_handle(self._pre_handles[1 if with_placeholder else 0]), self.n_hidden)
        else:
            fx = self._preprocess_eager(inp, with_placeholder)
        if T is not None:
            emb = timestep_embedding(T, self.n_hidden).repeat(1, x.shape[1], 1)
            fx = fx + self.time_fc(emb)
        for block in self.blocks:
            if getattr(self, "use_checkpoint", False) and torch.is_grad_enabled():
                fx = checkpoint.checkpoint(block, fx, use_reentrant=False)
            else:
                fx = block(fx)
        return fx


def reference_grid_distances(x, ref, dims):
    """Distances from every point to a ref^dims reference grid on [0,1]^dims
    (get_grid, HAETransolver_Irregular_Mesh.py:297-336); on x's device instead of the
    reference's hard-coded .cuda()."""
    grid = _reference_grid(ref, dims, x.device) if not torch.compiler.is_compiling() else _make_reference_grid(ref, dims, x.device)
    return torch.sqrt(torch.sum((x[:, :, None, :] - grid[:, None, :, :]) ** 2, dim=-1)).contiguous()


def _make_reference_grid(ref, dims, device):
    axes = [torch.tensor(np.linspace(0, 1, ref), dtype=torch.float, device=device)] * dims
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(1, ref ** dims, dims)


_GRIDS = {}


def _reference_grid(ref, dims, device):
    """The constant grid, built once per (ref, dims, device): rebuilding it was a host -> device copy and five tiny launches
    per forward of every `unified_pos` model (darcy, Navier-Stokes)."""
    key = (int(ref), int(dims), str(device))
    g = _GRIDS.get(key)
    if g is None:
        g = _GRIDS[key] = _make_reference_grid(ref, dims, device)
    return g
