"""Normalisation modules with the reference's names and parameters (layers/norm.py:7-88 there;
nn.LayerNorm arrives through `config.norm`, layers/vit.py:15,20).  Forward/backward run on the
wavefront-per-row kernels of csrc/norm.hip."""
import math

import torch
import torch.nn as nn

from .. import ops


class RMSNorm(nn.Module):
	"""Gemma-style RMSNorm: fp32 internally, y = x * rsqrt(mean(x^2) + eps) * (1 + w) when
	add_unit_offset (else * w), result in the dtype of x.  `weight` starts at zero."""

	def __init__(self, dim: int, eps: float = 1e-6, add_unit_offset: bool = True):
		super().__init__()
		self.eps = eps
		self.add_unit_offset = add_unit_offset
		self.weight = nn.Parameter(torch.zeros(dim))

	def forward(self, x):
		return ops.NormFn.apply(x, self.weight, None, "rmsnorm", self.eps, self.add_unit_offset)


class LayerNorm(nn.LayerNorm):
	"""nn.LayerNorm (same parameters and state_dict keys) whose forward runs on the vitk kernel.
	Use as `config.norm = LayerNorm`; a plain nn.LayerNorm inside ViTBlock/CvTBlock is routed to the
	same kernel by the block."""

	def forward(self, x):
		if len(self.normalized_shape) != 1 or self.weight is None or self.bias is None:
			raise NotImplementedError("vit_amd LayerNorm normalises the last dimension with affine weight and bias")
		return ops.NormFn.apply(x, self.weight, self.bias, "layernorm", self.eps, False)


def apply_norm(mod, x):
	"""Run `mod` (RMSNorm, LayerNorm, DyT or a stock nn.LayerNorm built by config.norm) on the vitk kernels."""
	if isinstance(mod, DyT):
		return mod(x)
	kind, w, b, eps, off = norm_spec(mod)
	return ops.NormFn.apply(x, w, b, kind, eps, off)


def norm_spec(mod):
	"""(kind, weight, bias, eps, unit_offset) of a supported norm module."""
	if isinstance(mod, RMSNorm):
		return "rmsnorm", mod.weight, None, mod.eps, mod.add_unit_offset
	if isinstance(mod, nn.LayerNorm):
		if len(mod.normalized_shape) != 1 or mod.weight is None or mod.bias is None:
			raise NotImplementedError("LayerNorm over the last dimension with affine parameters only")
		return "layernorm", mod.weight, mod.bias, mod.eps, False
	raise NotImplementedError(f"config.norm={type(mod).__name__} has no fused-block form (LayerNorm and RMSNorm have; DyT runs module by module)")


class DyT(nn.Module):
	"""Dynamic tanh 'norm' (layers/norm.py:35-45 of the reference): gamma * tanh(alpha * x) + beta with a learned scalar
	alpha, one bandwidth-bound kernel each way.  The result has the dtype of x (the reference's float32 parameters promote
	it to float32; the Linear that consumes it under autocast rounds it to bfloat16 either way)."""

	def __init__(self, C, init_alpha=0.5):
		super().__init__()
		self.alpha = nn.Parameter(torch.ones(1) * init_alpha)
		self.gamma = nn.Parameter(torch.ones(C))
		self.beta = nn.Parameter(torch.zeros(C))

	def forward(self, x):
		return ops.DyTFn.apply(x, self.alpha, self.gamma, self.beta)


class QKNorm(nn.Module):
	"""L2 query/key normalisation with a learned scalar gain initialised to
	sqrt(log2(L^2 - L)).  Import-compatible only: the reference's MHA cannot construct it
	(it passes d_head alone), so use_qk_norm=True is rejected by MHA here as well."""

	def __init__(self, d_head: int, max_seq_len: int, eps: float = 1e-6):
		super().__init__()
		self.eps = eps
		val = float(max_seq_len * max_seq_len - max_seq_len)
		if val < 1.0:
			val = 2.0
		self.gain = nn.Parameter(torch.tensor(math.sqrt(math.log2(val))))

	def forward(self, x):
		raise NotImplementedError("QKNorm has no MI355X kernel in this build")
