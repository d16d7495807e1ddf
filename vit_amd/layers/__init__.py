from .norm import DyT, LayerNorm, QKNorm, RMSNorm  # noqa: F401
from .transformers import FeedForward  # noqa: F401
from .attention import MHA, ConvAttention, QuadrangleAttention  # noqa: F401
from .embeddings import ConvEmbedding, patchify  # noqa: F401
from .vit import ViTBlock  # noqa: F401
from .cvt import CvTBlock  # noqa: F401
