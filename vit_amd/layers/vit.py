"""ViTBlock with the reference's interface (layers/vit.py:8-57 there): pre-norm attention and
feed-forward branches with optional LayerScale, returning (x, regularization_loss).

When nothing stochastic sits between the kernels (dropout inactive) the whole block is ONE autograd
node (ops.ViTBlockFn): 2 norm kernels, 4 GEMMs with fused bias / activation / residual epilogues and
the attention kernel in the forward."""
import torch
import torch.nn as nn

from .. import ops
from .attention import MHA, QuadrangleAttention
from .dropout import Dropout
from .norm import apply_norm, norm_spec
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

	def _fusable(self):
		return isinstance(self.attn, MHA) and not (self.attn_dropout.active() or self.ff_dropout.active()
			or self.ff.dropout1.active() or self.ff.dropout2.active() or self.attn.dropout.active())

	def forward(self, x, h=None, w=None):
		if self._fusable():
			kind, n1w, n1b, eps, off = norm_spec(self.attn_norm)
			kind2, n2w, n2b, eps2, off2 = norm_spec(self.ff_norm)
			assert (kind, eps, off) == (kind2, eps2, off2)
			cfg = (kind, eps, off, self.attn.n_head, self.ff.act_code, self.ff.gated)
			y = ops.ViTBlockFn.apply(x, n1w, n1b, self.attn.qkv_proj.weight, self.attn.out_proj.weight, self.attn.out_proj.bias,
				self.attn_layer_scale, n2w, n2b, self.ff.fc_in.weight, self.ff.fc_in.bias, self.ff.fc_out.weight, self.ff.fc_out.bias,
				self.ff_layer_scale, cfg)
			return y, 0.0
		# unfused composition (active dropout, or a user-supplied attention module)
		branch = self.attn_dropout(self.attn(apply_norm(self.attn_norm, x)))
		x = ops.ScaleResidualFn.apply(branch, x, self.attn_layer_scale)
		branch = self.ff_dropout(self.ff(apply_norm(self.ff_norm, x)))
		x = ops.ScaleResidualFn.apply(branch, x, self.ff_layer_scale)
		return x, 0.0
