from .erwin_flash import ErwinTransformer

__all__ = ["ErwinTransformer"]
