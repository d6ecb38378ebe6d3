"""HAETransolver_Structured_Mesh_3D.Model -- drop-in for
models/HAETransolver_Structured_Mesh_3D.py (block :27-149, Model :152-461)."""
import numpy as np
import torch

from ._model_base import ModelBase, TransolverErwinBlockBase
from .PhysicsAttention import Physics_Attention_Structured_Mesh_3D


class TransolverErwinBlock(TransolverErwinBlockBase):
    def __init__(self, num_heads, hidden_dim, dropout, act="gelu", mlp_ratio=4, last_layer=False,
                 out_dim=1, slice_num=32, H=32, W=32, D=32, c_hidden=None, ball_sizes=None,
                 enc_num_heads=None, enc_depths=None, dec_num_heads=None, dec_depths=None, strides=None,
                 rotate=45, decode=True, mp_steps=0, embed=False):
        super().__init__()
        attn = Physics_Attention_Structured_Mesh_3D(
            hidden_dim, heads=num_heads, dim_head=hidden_dim // num_heads, dropout=dropout,
            slice_num=slice_num, H=H, W=W, D=D, c_hidden=c_hidden, ball_sizes=ball_sizes,
            enc_num_heads=enc_num_heads, enc_depths=enc_depths, dec_num_heads=dec_num_heads,
            dec_depths=dec_depths, strides=strides, rotate=rotate, decode=decode, mlp_ratio=mlp_ratio,
            mp_steps=mp_steps, embed=embed)
        self._init_block(attn, hidden_dim, act, mlp_ratio, last_layer, out_dim)


class Model(ModelBase):
    def __init__(self, space_dim=1, n_layers=5, n_hidden=256, dropout=0.0, n_head=8, Time_Input=False,
                 act="gelu", mlp_ratio=1, fun_dim=1, out_dim=1, slice_num=32, ref=8, unified_pos=False,
                 H=32, W=32, D=32, c_hidden=None, ball_sizes=None, enc_num_heads=None, enc_depths=None,
                 dec_num_heads=None, dec_depths=None, strides=None, rotate=45, decode=True, mp_steps=0,
                 embed=False):
        super().__init__()
        self.__name__ = "Transolver_3D"
        self.use_checkpoint = False  # (HAETransolver_Structured_Mesh_3D.py:243; set True for gradient checkpointing)
        self.H, self.W, self.D, self.ref = H, W, D, ref
        self.unified_pos, self.Time_Input = unified_pos, Time_Input
        self.n_hidden, self.space_dim = n_hidden, space_dim
        if unified_pos:
            self.pos = self.get_grid()
        in_dim = fun_dim + (ref ** 3 if unified_pos else space_dim)

        def make_block(last):
            return TransolverErwinBlock(
                num_heads=n_head, hidden_dim=n_hidden, dropout=dropout, act=act, mlp_ratio=mlp_ratio,
                out_dim=out_dim, slice_num=slice_num, H=H, W=W, D=D, last_layer=last, c_hidden=c_hidden,
                ball_sizes=ball_sizes, enc_num_heads=enc_num_heads, enc_depths=enc_depths,
                dec_num_heads=dec_num_heads, dec_depths=dec_depths, strides=strides, rotate=rotate,
                decode=decode, mp_steps=mp_steps, embed=embed)

        self._init_model(make_block, in_dim, n_layers, n_hidden, act, Time_Input)

    def get_grid(self, batchsize=1):
        def axis(n):
            return torch.tensor(np.linspace(0, 1, n), dtype=torch.float)

        g = torch.stack(torch.meshgrid(axis(self.H), axis(self.W), axis(self.D), indexing="ij"), dim=-1)
        r = torch.stack(torch.meshgrid(axis(self.ref), axis(self.ref), axis(self.ref), indexing="ij"),
                        dim=-1).reshape(self.ref ** 3, 3)
        pos = torch.sqrt(((g[:, :, :, None, :] - r[None, None, None, :, :]) ** 2).sum(-1))
        return pos.reshape(1, self.H, self.W, self.D, self.ref ** 3).contiguous()

    def unified_positions(self, x):
        return self.pos.to(x.device).repeat(x.shape[0], 1, 1, 1, 1).reshape(
            x.shape[0], self.H * self.W * self.D, self.ref ** 3)
