"""ViT glue with the reference's constructor, attributes and state_dict keys (models/vit.py:12-69 there):
`class_token [1,d]`, `pos_embed [1,N,d]`, `linear_mapper`, `blocks.{i}.*`, `mlp`.

The reference's forward materialises patches, loops over the batch in Python to prepend the class
token and repeats pos_embed; here the patch GEMM gathers pixels in its A-tile loader, adds bias and
pos_embed in its epilogue and writes straight into rows 1.. of the token buffer, a small kernel fills
row 0, and the classifier reads row 0 in place through the row stride."""
import os

import torch
import torch.nn as nn

from .. import ops
from ..layers.attention import QuadrangleAttention
from ..layers.vit import ViTBlock


CLASS_ROWS_LAST_BLOCK = os.environ.get("VITK_CLASS_ROWS_LAST_BLOCK", "1") != "0"  # see ViT.forward


class ViT(nn.Module):
	def __init__(self, config):
		super().__init__()
		self.chw = config.chw
		self.d_model = config.d_model
		self.n_patch = config.n_patch
		self.n_block = config.n_block
		c, h, w = self.chw
		assert h % self.n_patch == 0, "Input shape not entirely divisible by number of patch"
		assert w % self.n_patch == 0, "Input shape not entirely divisible by number of patch"
		self.patch_size = (h // self.n_patch, w // self.n_patch)
		self.h_grid = h // self.patch_size[0]
		self.w_grid = w // self.patch_size[1]
		self.input_d = c * self.patch_size[0] * self.patch_size[1]

		self.linear_mapper = nn.Linear(self.input_d, self.d_model)
		self.class_token = nn.Parameter(torch.rand(1, self.d_model))
		self.pos_embed = nn.Parameter(torch.randn(1, self.n_patch ** 2 + 1, self.d_model))
		self.blocks = nn.ModuleList([ViTBlock(config) for _ in range(self.n_block)])
		self.mlp = nn.Linear(self.d_model, config.out_d)
		self.is_quadrangle_attention = config.attention == QuadrangleAttention

	def forward(self, images):
		x = ops.PatchEmbedFn.apply(images, self.linear_mapper.weight, self.linear_mapper.bias, self.class_token, self.pos_embed, self.n_patch)
		n = len(self.blocks)
		for i, block in enumerate(self.blocks):
			# the head reads out[:, 0] only (models/vit.py:63 there): the last block computes the rows that are read
			last = block.forward_class_rows(x) if (i == n - 1 and CLASS_ROWS_LAST_BLOCK) else None
			x, _ = last if last is not None else block(x)
		return ops.ClsHeadFn.apply(x, self.mlp.weight, self.mlp.bias)
