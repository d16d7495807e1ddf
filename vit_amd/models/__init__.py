from .vit import ViT  # noqa: F401
from .cvt import CvT  # noqa: F401
from .deit import DeiTViT  # noqa: F401
from .coatnet import CoAtNet  # noqa: F401
