from .vit import ViT  # noqa: F401
from .cvt import CvT  # noqa: F401
