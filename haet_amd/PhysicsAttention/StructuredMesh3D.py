"""Physics_Attention_Structured_Mesh_3D -- drop-in for models/PhysicsAttention/
StructuredMesh3D.py (constructor :45-71, forward :161-266).  The 3x3x3 Conv3d projection is
an implicit GEMM on the matrix cores (haet_conv_project_fwd; MIOpen only for channel counts it does not
take); everything after it is the shared HIP path."""
import torch.nn as nn

from ._base import PhysicsAttentionBase


class Physics_Attention_Structured_Mesh_3D(PhysicsAttentionBase):
    dimensionality = 3  # StructuredMesh3D.py:106

    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0, slice_num=32, H=32, W=32, D=32,
                 kernel=3, base_temp=0.5, epsilon=1e-6, c_hidden=None, ball_sizes=None,
                 enc_num_heads=None, enc_depths=None, dec_num_heads=None, dec_depths=None,
                 strides=None, rotate=1, decode=True, mlp_ratio=4, mp_steps=0, embed=False):
        super().__init__()
        self.H, self.W, self.D = H, W, D
        self.in_project_x = nn.Conv3d(dim, dim_head * heads, kernel, 1, kernel // 2)
        self._init_common(dim, heads, dim_head, dropout, slice_num, base_temp, epsilon, c_hidden,
                          ball_sizes, enc_num_heads, enc_depths, dec_num_heads, dec_depths, strides,
                          rotate, decode, mlp_ratio, mp_steps, embed)

    def project_torch(self, x):
        B, N, Dm = x.shape
        g = x.reshape(B, self.H, self.W, self.D, Dm).permute(0, 4, 1, 2, 3).contiguous()
        return self.in_project_x(g).permute(0, 2, 3, 4, 1).reshape(B, N, -1)

    def _prime_points(self):
        return self.H * self.W * self.D

    def grid_shape(self):
        return (self.H, self.W, self.D)

    def project_split(self, x, sp, pre_norm=None):
        # A LayerNorm in front cannot ride in the convolution's own loader (the taps read NEIGHBOUR rows); it rides in the
        # pre-split pass of the LDS-DMA convolution instead (ops.conv_presplit), where that kernel applies: fuses_pre_norm
        if x.dtype != __import__("torch").float32 or (pre_norm is not None and not self.fuses_pre_norm(x, pre_norm)):
            return None
        return self._conv_project_split(x, (self.H, self.W, self.D), sp, pre_norm)

    def fuses_pre_norm(self, x, ln):
        return self._conv_fuses_pre_norm(x, ln)

    def project(self, x):
        B, N, Dm = x.shape
        y = self._conv_project(x, (self.H, self.W, self.D))  # HIP implicit GEMM, channels-last
        if y is not None:
            return y
        g = x.reshape(B, self.H, self.W, self.D, Dm).permute(0, 4, 1, 2, 3).contiguous()
        return self.in_project_x(g).permute(0, 2, 3, 4, 1).reshape(B, N, -1)
