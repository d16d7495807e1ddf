"""A DeiT-style student on the ViT of models/vit.py (there): one more learned token (the distillation token, row 1 of
every image) and one more Linear head that reads it.  The reference has no such model: its DeiTTrainer distils the
class-token logits themselves (deit_trainer.py:30-45 there; only an old trainer snapshot accepts a (prediction,
distillation) pair, _single_file.py:909-910).  This is the build-side student the fused heads and the dual loss serve;
`pos_embed` has n_patch^2 + 2 rows, everything else keeps the ViT's parameter names."""
import torch
import torch.nn as nn

from .. import ops
from .vit import ViT


class DeiTViT(ViT):
	def __init__(self, config):
		super().__init__(config)
		self.dist_token = nn.Parameter(torch.rand(1, self.d_model))
		self.pos_embed = nn.Parameter(torch.randn(1, self.n_patch ** 2 + 2, self.d_model))
		self.dist_head = nn.Linear(self.d_model, config.out_d)

	def forward(self, images):
		"""Training: (class logits, distillation logits).  Evaluation: their mean, the DeiT inference rule."""
		x = ops.PatchEmbedFn.apply(images, self.linear_mapper.weight, self.linear_mapper.bias, self.class_token, self.pos_embed, self.n_patch, self.dist_token)
		for block in self.blocks:
			x, _ = block(x)
		cls, dist = ops.DualHeadFn.apply(x, self.mlp.weight, self.mlp.bias, self.dist_head.weight, self.dist_head.bias)
		if self.training:
			return cls, dist
		return (cls.float() + dist.float()) * 0.5
