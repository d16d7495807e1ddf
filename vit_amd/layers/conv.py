"""MBConv with the reference's interface (layers/conv.py:5-44 there): the inverted-bottleneck convolution block of its
CoAtNet.  Parameters live in the same nn.Sequential (`conv.0` ... `conv.7`), so reference state_dicts load unchanged;
the computation is ops.MBConvFn: two MFMA GEMMs (the 1x1 convolutions), the token-major depthwise 3x3 kernel and three
token-major BatchNorm kernels."""
import torch.nn as nn

from .. import _lib as L
from .. import ops


class MBConv(nn.Module):
	def __init__(self, in_chans, out_chans, stride=1, expand_ratio=4, activation="gelu"):
		super().__init__()
		self.stride = stride
		hidden_dim = in_chans * expand_ratio
		self.activation = activation.lower()
		# The reference selects the activation with `if gelu ... if silu ... else ReLU` (layers/conv.py:12-17): without an
		# `elif`, "gelu" falls through to the else branch, so only "silu" survives and everything else is ReLU.  Kept.
		self.act_fn = nn.SiLU if self.activation == "silu" else nn.ReLU
		self.act_code = L.ACT_SILU if self.activation == "silu" else L.ACT_RELU
		self.use_res_connect = self.stride == 1 and in_chans == out_chans
		self.conv = nn.Sequential(
			nn.Conv2d(in_chans, hidden_dim, 1, 1, 0, bias=False), nn.BatchNorm2d(hidden_dim), self.act_fn(),
			nn.Conv2d(hidden_dim, hidden_dim, 3, stride, 1, groups=hidden_dim, bias=False), nn.BatchNorm2d(hidden_dim), self.act_fn(),
			nn.Conv2d(hidden_dim, out_chans, 1, 1, 0, bias=False), nn.BatchNorm2d(out_chans))

	def forward(self, x, hw=None):
		"""x: NCHW as in the reference (NCHW out), or token-major [B, H*W, C] with hw = (H, W) (token-major out): what the
		conv + attention hybrid passes from block to block, so that no layout conversion runs between them."""
		c = self.conv
		bns = (c[1], c[4], c[7])
		assert all(b.eps == bns[0].eps and b.momentum == bns[0].momentum for b in bns)
		if self.training:
			for b in bns:
				b.num_batches_tracked += 1
		cfg = (self.stride, self.act_code, self.use_res_connect, bns[0].eps, bns[0].momentum, self.training, hw)
		return ops.MBConvFn.apply(x, c[0].weight, c[1].weight, c[1].bias, c[1].running_mean, c[1].running_var,
			c[3].weight, c[4].weight, c[4].bias, c[4].running_mean, c[4].running_var,
			c[6].weight, c[7].weight, c[7].bias, c[7].running_mean, c[7].running_var, cfg)
