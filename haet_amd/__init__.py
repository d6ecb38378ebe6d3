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


def _freeze_import_heap():
    """Importing this package imports torch and, through haet_amd._compile, torch._dynamo with its symbolic-math stack:
    ~10^6 long-lived objects that every FULL garbage collection walks again -- ~75 ms per collection on the GPU boxes
    (tools/stall_probe.py), i.e. two or three training steps of the reference's PDE configurations or fifty 1M-point
    forwards, at moments the caller does not choose.  gc.freeze() moves what is alive NOW (import-time objects) out of the
    collector's reach; the collector stays on for everything created afterwards.  HAET_NO_GC_FREEZE=1 leaves it alone."""
    import gc
    import os

    if not os.environ.get("HAET_NO_GC_FREEZE"):
        gc.collect()
        gc.freeze()


_freeze_import_heap()
