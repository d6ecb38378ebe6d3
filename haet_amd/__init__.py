"""haet_amd -- MI355X-native PhysicsAttention hot path of pedrocurvo/HAET.

Mirrors the reference's `models` package surface (models/__init__.py:15-17):

    from haet_amd import HAETransolver_Irregular_Mesh, HAETransolver_Structured_Mesh_2D, \
        HAETransolver_Structured_Mesh_3D
    from haet_amd.PhysicsAttention import Physics_Attention_Irregular_Mesh, ...

so `import haet_amd as models` drops into the reference's training scripts.
The compute path is libhaet_hip.so (hand-written HIP for gfx950); see include/haet_hip.h.
"""
from ._lib import precision, set_precision  # noqa: F401  (arithmetic switch: "bf16x3" default, "fp16x2" opt-in)
from .HAETransolver_Irregular_Mesh import Model as HAETransolver_Irregular_Mesh
from .HAETransolver_Structured_Mesh_2D import Model as HAETransolver_Structured_Mesh_2D
from .HAETransolver_Structured_Mesh_3D import Model as HAETransolver_Structured_Mesh_3D

__all__ = [
    "HAETransolver_Irregular_Mesh",
    "HAETransolver_Structured_Mesh_2D",
    "HAETransolver_Structured_Mesh_3D",
]
