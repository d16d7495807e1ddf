"""Attention modules with the reference's interface (layers/attention.py there).

MHA: fused QKV GEMM -> head-strided attention kernel (reads q/k/v as column slices of the QKV
buffer, writes the merged-head layout) -> output-projection GEMM.  None of the reference's five
rearrange copies is materialised.
ConvAttention (CvT): q/k/v GEMMs, depthwise 3x3 convolution of K and V directly on token-major
activations, attention with Skv != Sq, output projection.
"""
import torch
import torch.nn as nn

from .. import ops
from .dropout import Dropout
from .norm import QKNorm  # noqa: F401  (import parity with the reference module)


class MHA(nn.Module):
	def __init__(self, config):
		super().__init__()
		self.d_model = config.d_model
		self.n_head = config.n_head
		self.d_head = self.d_model // self.n_head
		if self.d_head * self.n_head != self.d_model:
			raise ValueError("d_model must be divisible by n_head")
		self.qkv_proj = nn.Linear(self.d_model, 3 * self.d_model, bias=False)
		self.out_proj = nn.Linear(self.d_model, self.d_model)
		self.dropout = Dropout(config.dropout)
		self.use_qk_norm = getattr(config, "use_qk_norm", False)
		if self.use_qk_norm:
			raise NotImplementedError("use_qk_norm=True is not supported (the reference's QKNorm cannot be constructed by its MHA either)")

	def forward(self, query):
		# dropout_p=self.dropout.p if self.training else 0.0 (layers/attention.py:41 there)
		drop = (self.dropout.p, ops.next_dropout_seed()) if self.dropout.active() else (0.0, 0)
		return ops.MHAFn.apply(query, self.qkv_proj.weight, self.out_proj.weight, self.out_proj.bias, self.n_head, drop)


class ConvAttention(nn.Module):
	def __init__(self, dim, n_head, kernel_size=3, qkv_stride=1, padding=1, dropout=0.1):
		super().__init__()
		self.n_head = n_head
		self.dim = dim
		self.d_head = dim // n_head
		self.q_proj = nn.Linear(dim, dim)
		self.k_proj = nn.Linear(dim, dim)
		self.v_proj = nn.Linear(dim, dim)
		self.conv_proj_k = nn.Conv2d(dim, dim, kernel_size, stride=qkv_stride, padding=padding, groups=dim, bias=False)
		self.conv_proj_v = nn.Conv2d(dim, dim, kernel_size, stride=qkv_stride, padding=padding, groups=dim, bias=False)
		self.out_proj = nn.Linear(dim, dim)
		self.dropout = Dropout(dropout)
		self.kernel_size, self.qkv_stride, self.padding = kernel_size, qkv_stride, padding

	def forward(self, x, h, w):
		# dropout_p=self.dropout.p if self.training else 0.0 (layers/attention.py:86 there)
		drop = (self.dropout.p, ops.next_dropout_seed()) if self.dropout.active() else (0.0, 0)
		return ops.ConvAttentionFn.apply(x, self.q_proj.weight, self.q_proj.bias, self.k_proj.weight, self.k_proj.bias,
			self.v_proj.weight, self.v_proj.bias, self.conv_proj_k.weight, self.conv_proj_v.weight,
			self.out_proj.weight, self.out_proj.bias, (h, w, self.n_head, self.kernel_size, self.qkv_stride, self.padding, drop))


class QuadrangleAttention(nn.Module):
	"""Learned-quadrangle window attention of the reference (layers/attention.py:95-283 there).
	Out of scope of the MI355X hot path; the name exists so that configs and `isinstance` checks
	written against the reference import cleanly.  Select `attention = MHA` in the config."""

	def __init__(self, config, lambda_reg=1.0):
		super().__init__()
		raise NotImplementedError("QuadrangleAttention is not part of the MI355X hot path: set config.attention = MHA")
