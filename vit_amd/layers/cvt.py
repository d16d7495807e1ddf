"""CvTBlock with the reference's interface (layers/cvt.py:8-50 there): the ViT block skeleton around
ConvAttention, returning (x, h, w)."""
import torch
import torch.nn as nn

from .. import ops
from .attention import ConvAttention
from .dropout import Dropout
from .norm import DyT, apply_norm, norm_spec
from .transformers import FeedForward

FUSED_BLOCK = True  # one autograd node per block (ops.CvTBlockFn) where the block allows it; False: always module by module


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

	def _fusable(self):
		if not FUSED_BLOCK or isinstance(self.attn_norm, DyT) or isinstance(self.ff_norm, DyT):
			return False
		drops = (self.attn.dropout, self.attn_dropout, self.ff.dropout1, self.ff.dropout2, self.ff_dropout)
		return not any(m.active() for m in drops)

	def forward(self, x, h, w):
		if self._fusable():
			kind, n1w, n1b, eps, off = norm_spec(self.attn_norm)
			kind2, n2w, n2b, eps2, off2 = norm_spec(self.ff_norm)
			assert (kind, eps, off) == (kind2, eps2, off2)
			at = self.attn
			cfg = (kind, eps, off, h, w, at.n_head, at.kernel_size, at.qkv_stride, at.padding, self.ff.act_code, self.ff.gated)
			y = ops.CvTBlockFn.apply(x, n1w, n1b, at.q_proj.weight, at.q_proj.bias, at.k_proj.weight, at.k_proj.bias, at.v_proj.weight, at.v_proj.bias,
				at.conv_proj_k.weight, at.conv_proj_v.weight, at.out_proj.weight, at.out_proj.bias, self.attn_layer_scale,
				n2w, n2b, self.ff.fc_in.weight, self.ff.fc_in.bias, self.ff.fc_out.weight, self.ff.fc_out.bias, self.ff_layer_scale, cfg)
			return y, h, w
		# module-by-module composition (an active dropout, or DyT in place of a norm)
		branch = self.attn_dropout(self.attn(apply_norm(self.attn_norm, x), h, w))
		x = ops.ScaleResidualFn.apply(branch, x, self.attn_layer_scale)
		branch = self.ff_dropout(self.ff(apply_norm(self.ff_norm, x)))
		x = ops.ScaleResidualFn.apply(branch, x, self.ff_layer_scale)
		return x, h, w
