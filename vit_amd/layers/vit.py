"""ViTBlock with the reference's interface (layers/vit.py:8-57 there): pre-norm attention and
feed-forward branches with optional LayerScale, returning (x, regularization_loss).

The whole block is ONE autograd node (ops.ViTBlockFn): 2 norm kernels, 4 GEMMs with fused bias / activation /
residual epilogues and the attention kernel in the forward.  Active dropout (training mode) stays inside that node:
the attention kernel drops probabilities itself, FeedForward.dropout1 is applied in place on fc_out's input, and each
branch dropout runs fused with its residual add (masks are regenerated in the backward, never stored)."""
import torch
import torch.nn as nn

from .. import ops
from .attention import MHA, QuadrangleAttention
from .dropout import Dropout
from .norm import DyT, apply_norm, norm_spec
from .transformers import FeedForward


class ViTBlock(nn.Module):
	def __init__(self, config):
		super().__init__()
		self.d_model = config.d_model
		self.n_head = config.n_head
		use_ls = config.use_layer_scale

		self.attn_norm = config.norm(self.d_model)
		self.attn = config.attention(config)
		self.attn_layer_scale = nn.Parameter(torch.full((self.d_model,), 1e-4)) if use_ls else None
		self.attn_dropout = Dropout(config.dropout)

		self.ff_norm = config.norm(self.d_model)
		self.ff = FeedForward(config)
		self.ff_layer_scale = nn.Parameter(torch.full((self.d_model,), 1e-4)) if use_ls else None
		self.ff_dropout = Dropout(config.dropout)

		self.is_quadrangle_attention = isinstance(self.attn, QuadrangleAttention)

	def _dropout_slots(self):
		return (("attn", self.attn.dropout), ("a", self.attn_dropout), ("f1", self.ff.dropout1), ("f2", self.ff.dropout2), ("b", self.ff_dropout))

	def _fusable(self):
		# with active dropout the fused node covers the block without LayerScale (the reference's default)
		if not isinstance(self.attn, MHA) or isinstance(self.attn_norm, DyT) or isinstance(self.ff_norm, DyT):
			return False
		if self.attn_layer_scale is None and self.ff_layer_scale is None:
			return True
		return not any(m.active() for _, m in self._dropout_slots())

	def forward_class_rows(self, x):
		"""The block as the LAST block of a model whose head reads token row 0 only: returns ([B, 1, d], 0.0), the rows
		of the block's output that are read (ops.ViTLastBlockFn), or None when this block cannot take that form (LayerScale,
		an active dropout, another attention or norm module) and the caller should run forward()."""
		if not self._fusable() or self.attn_layer_scale is not None or self.ff_layer_scale is not None or any(m.active() for _, m in self._dropout_slots()):
			return None
		kind, n1w, n1b, eps, off = norm_spec(self.attn_norm)
		kind2, n2w, n2b, eps2, off2 = norm_spec(self.ff_norm)
		assert (kind, eps, off) == (kind2, eps2, off2)
		cfg = (kind, eps, off, self.attn.n_head, self.ff.act_code, self.ff.gated, None)
		y = ops.ViTLastBlockFn.apply(x, n1w, n1b, self.attn.qkv_proj.weight, self.attn.out_proj.weight, self.attn.out_proj.bias,
			n2w, n2b, self.ff.fc_in.weight, self.ff.fc_in.bias, self.ff.fc_out.weight, self.ff.fc_out.bias, cfg)
		return y, 0.0

	def forward(self, x, h=None, w=None):
		if self._fusable():
			# {slot: (p, seed)} of the dropouts that are on right now (training mode, p > 0); fresh seeds per call, drawn
			# in the order the module-by-module composition below would draw them
			drops = {k: (m.p, ops.next_dropout_seed()) for k, m in self._dropout_slots() if m.active()}
			kind, n1w, n1b, eps, off = norm_spec(self.attn_norm)
			kind2, n2w, n2b, eps2, off2 = norm_spec(self.ff_norm)
			assert (kind, eps, off) == (kind2, eps2, off2)
			cfg = (kind, eps, off, self.attn.n_head, self.ff.act_code, self.ff.gated, drops or None)
			y = ops.ViTBlockFn.apply(x, n1w, n1b, self.attn.qkv_proj.weight, self.attn.out_proj.weight, self.attn.out_proj.bias,
				self.attn_layer_scale, n2w, n2b, self.ff.fc_in.weight, self.ff.fc_in.bias, self.ff.fc_out.weight, self.ff.fc_out.bias,
				self.ff_layer_scale, cfg)
			return y, 0.0
		# module-by-module composition (LayerScale together with active dropout, or a user-supplied attention module)
		branch = self.attn_dropout(self.attn(apply_norm(self.attn_norm, x)))
		x = ops.ScaleResidualFn.apply(branch, x, self.attn_layer_scale)
		branch = self.ff_dropout(self.ff(apply_norm(self.ff_norm, x)))
		x = ops.ScaleResidualFn.apply(branch, x, self.ff_layer_scale)
		return x, 0.0
