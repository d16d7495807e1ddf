"""CvTBlock with the reference's interface (layers/cvt.py:8-50 there): the ViT block skeleton around
ConvAttention, returning (x, h, w)."""
import torch
import torch.nn as nn

from .. import ops
from .attention import ConvAttention
from .dropout import Dropout
from .norm import apply_norm
from .transformers import FeedForward


class CvTBlock(nn.Module):
	def __init__(self, config, kernel_size, qkv_stride, padding):
		super().__init__()
		self.d_model = config.d_model
		use_ls = config.use_layer_scale
		self.attn_norm = config.norm(self.d_model)
		self.attn = ConvAttention(dim=self.d_model, n_head=config.n_head, kernel_size=kernel_size, qkv_stride=qkv_stride,
			padding=padding, dropout=config.dropout)
		self.attn_layer_scale = nn.Parameter(torch.full((self.d_model,), 1e-4)) if use_ls else None
		self.attn_dropout = Dropout(config.dropout)
		self.ff_norm = config.norm(self.d_model)
		self.ff = FeedForward(config)
		self.ff_layer_scale = nn.Parameter(torch.full((self.d_model,), 1e-4)) if use_ls else None
		self.ff_dropout = Dropout(config.dropout)

	def forward(self, x, h, w):
		branch = self.attn_dropout(self.attn(apply_norm(self.attn_norm, x), h, w))
		x = ops.ScaleResidualFn.apply(branch, x, self.attn_layer_scale)
		branch = self.ff_dropout(self.ff(apply_norm(self.ff_norm, x)))
		x = ops.ScaleResidualFn.apply(branch, x, self.ff_layer_scale)
		return x, h, w
