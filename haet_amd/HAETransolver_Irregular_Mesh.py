"""HAETransolver_Irregular_Mesh.Model -- drop-in for models/HAETransolver_Irregular_Mesh.py
(TransolverErwinBlock :26-134, Model :137-376)."""
from ._model_base import ModelBase, TransolverErwinBlockBase, reference_grid_distances
from .PhysicsAttention import Physics_Attention_Irregular_Mesh


class TransolverErwinBlock(TransolverErwinBlockBase):
    def __init__(self, num_heads, hidden_dim, dropout, act="gelu", mlp_ratio=4, last_layer=False,
                 out_dim=1, slice_num=32, c_hidden=None, ball_sizes=None, enc_num_heads=None,
                 enc_depths=None, dec_num_heads=None, dec_depths=None, strides=None, rotate=45,
                 decode=True, mp_steps=0, embed=False):
        super().__init__()
        attn = Physics_Attention_Irregular_Mesh(
            hidden_dim, heads=num_heads, dim_head=hidden_dim // num_heads, dropout=dropout,
            slice_num=slice_num, c_hidden=c_hidden, ball_sizes=ball_sizes, enc_num_heads=enc_num_heads,
            enc_depths=enc_depths, dec_num_heads=dec_num_heads, dec_depths=dec_depths, strides=strides,
            rotate=rotate, decode=decode, mlp_ratio=mlp_ratio, mp_steps=mp_steps, embed=embed)
        self._init_block(attn, hidden_dim, act, mlp_ratio, last_layer, out_dim)


class Model(ModelBase):
    placeholder_always = True  # HAETransolver_Irregular_Mesh.py:364

    def __init__(self, space_dim=1, n_layers=5, n_hidden=256, dropout=0.0, n_head=8, Time_Input=False,
                 act="gelu", mlp_ratio=1, fun_dim=1, out_dim=1, slice_num=32, ref=8, unified_pos=False,
                 c_hidden=None, ball_sizes=None, enc_num_heads=None, enc_depths=None, dec_num_heads=None,
                 dec_depths=None, strides=None, rotate=45, decode=True, mp_steps=0, embed=False):
        super().__init__()
        self.__name__ = "Transolver_1D"
        self.ref, self.unified_pos, self.Time_Input = ref, unified_pos, Time_Input
        self.n_hidden, self.space_dim = n_hidden, space_dim
        in_dim = fun_dim + (ref * ref if unified_pos else space_dim)

        def make_block(last):
            return TransolverErwinBlock(
                num_heads=n_head, hidden_dim=n_hidden, dropout=dropout, act=act, mlp_ratio=mlp_ratio,
                out_dim=out_dim, slice_num=slice_num, last_layer=last, c_hidden=c_hidden,
                ball_sizes=ball_sizes, enc_num_heads=enc_num_heads, enc_depths=enc_depths,
                dec_num_heads=dec_num_heads, dec_depths=dec_depths, strides=strides, rotate=rotate,
                decode=decode, mp_steps=mp_steps, embed=embed)

        self._init_model(make_block, in_dim, n_layers, n_hidden, act, Time_Input)

    def get_grid(self, x, batchsize=1):
        return reference_grid_distances(x, self.ref, 2)

    def unified_positions(self, x):
        return self.get_grid(x, x.shape[0])
