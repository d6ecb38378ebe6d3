from .IrregularMesh import Physics_Attention_Irregular_Mesh
from .StructuredMesh2D import Physics_Attention_Structured_Mesh_2D
from .StructuredMesh3D import Physics_Attention_Structured_Mesh_3D

__all__ = [
    "Physics_Attention_Structured_Mesh_3D",
    "Physics_Attention_Structured_Mesh_2D",
    "Physics_Attention_Irregular_Mesh",
]
