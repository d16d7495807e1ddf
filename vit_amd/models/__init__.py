from .vit import ViT  # noqa: F401
