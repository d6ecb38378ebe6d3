"""Physics_Attention_Irregular_Mesh -- drop-in for models/PhysicsAttention/IrregularMesh.py
(constructor :41-63, forward :141-235, same state_dict keys)."""
import torch.nn as nn

from ._base import PhysicsAttentionBase


class Physics_Attention_Irregular_Mesh(PhysicsAttentionBase):
    dimensionality = 3  # hard-coded in the reference (IrregularMesh.py:91)

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0, slice_num=64, base_temp=0.5,
                 epsilon=1e-6, c_hidden=None, ball_sizes=None, enc_num_heads=None, enc_depths=None,
                 dec_num_heads=None, dec_depths=None, strides=None, rotate=1, decode=True,
                 mlp_ratio=4, mp_steps=0, embed=False):
        super().__init__()
        self.in_project_x = nn.Linear(dim, dim_head * heads)
        self._init_common(dim, heads, dim_head, dropout, slice_num, base_temp, epsilon, c_hidden,
                          ball_sizes, enc_num_heads, enc_depths, dec_num_heads, dec_depths, strides,
                          rotate, decode, mlp_ratio, mp_steps, embed)

    def project(self, x):
        # [B, N, H*C]; no [B,H,N,C] transpose is ever made
        return self._linear("in_project_x", self.in_project_x, x)

    def project_torch(self, x):
        return self.in_project_x(x)

    def project_backward(self, x, g_lin, need_x, wanted):
        lin = self.in_project_x
        from .. import ops

        grads = {}
        if id(lin.weight) in wanted or (lin.bias is not None and id(lin.bias) in wanted):
            grads[id(lin.weight)], gb = ops.linear_wgrad(g_lin, x)
            if lin.bias is not None:
                grads[id(lin.bias)] = gb
        return (self._linear_transposed("in_project_x", lin, g_lin) if need_x else None), grads

    def project_split(self, x, sp, pre_norm=None):
        return self._project_split_linear("in_project_x", self.in_project_x, x, sp, pre_norm)

    def fuses_pre_norm(self, x, ln):
        return (ln.elementwise_affine and ln.bias is not None and tuple(ln.normalized_shape) == (self.in_project_x.in_features,)
                and self.shard_group is None and self._split_linear_applies(self.in_project_x, x))
