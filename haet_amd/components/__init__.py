from .embedding import timestep_embedding
from .erwinflash import ErwinTransformer as ErwinFlashTransformer
from .mlp import MLP

__all__ = ["MLP", "ErwinFlashTransformer", "timestep_embedding"]
