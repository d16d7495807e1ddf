"""CvT glue with the reference's constructor, attributes and state_dict keys (models/cvt.py:8-108 there):
three stages of ConvEmbedding + CvTBlocks, token mean pool, norm, `head`.

Between stages the reference rearranges tokens back to NCHW for the next convolution; here the next
stage's im2col gather reads the token-major activations directly, so that round trip does not exist."""
import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..layers.cvt import CvTBlock
from ..layers.embeddings import ConvEmbedding
from ..layers.norm import apply_norm


class _StageConfig:
	pass


class CvT(nn.Module):
	def __init__(self, config):
		super().__init__()
		self.config = config
		in_ch = config.chw[0]
		for s in (1, 2, 3):
			dim = getattr(config, f"s{s}_emb_dim")
			emb = ConvEmbedding(in_channels=in_ch, out_channels=dim, kernel_size=getattr(config, f"s{s}_emb_kernel"),
				stride=getattr(config, f"s{s}_emb_stride"), padding=getattr(config, f"s{s}_emb_pad"))
			blocks = nn.ModuleList([
				CvTBlock(config=self.get_stage_config(dim, getattr(config, f"s{s}_heads"), getattr(config, f"s{s}_mlp_ratio")),
					kernel_size=getattr(config, f"s{s}_qkv_kernel"), qkv_stride=getattr(config, f"s{s}_qkv_stride"), padding=getattr(config, f"s{s}_qkv_pad"))
				for _ in range(getattr(config, f"s{s}_depth"))])
			setattr(self, f"stage{s}_embedding", emb)
			setattr(self, f"stage{s}_blocks", blocks)
			in_ch = dim
		self.norm = config.norm(config.s3_emb_dim)
		self.head = nn.Linear(config.s3_emb_dim, config.out_d)

	def get_stage_config(self, d_model, n_head, mlp_ratio):
		c = _StageConfig()
		c.d_model = d_model
		c.d_ff = int(d_model * mlp_ratio)
		c.n_head = n_head
		c.norm = self.config.norm
		c.activation = self.config.activation
		c.dropout = self.config.dropout
		c.use_layer_scale = self.config.use_layer_scale
		return c

	def forward(self, x):
		h, w = self.config.chw[1], self.config.chw[2]
		hw = None  # stage 1 reads the NCHW image; later stages read tokens
		for s in (1, 2, 3):
			x = getattr(self, f"stage{s}_embedding")(x, hw)
			stride = getattr(self.config, f"s{s}_emb_stride")
			h, w = h // stride, w // stride
			for blk in getattr(self, f"stage{s}_blocks"):
				x, h, w = blk(x, h, w)
			hw = (h, w)
		x = ops.MeanPoolFn.apply(x)
		x = apply_norm(self.norm, x)
		return ops.LinearFn.apply(x, self.head.weight, self.head.bias, L.ACT_NONE)
