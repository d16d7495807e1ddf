"""CoAtNet-style hybrid (conv stem, two MBConv stages, two transformer stages) on the kernel library: the build-side harness
for the large-batch 224x224 hybrid configuration.  The reference's HEAD class cannot be constructed (models/coatnet.py:11-91
there stops on undefined names); the structure here follows the constructible text of the older snapshot
(_single_file.py:399-458 there, configuration 562-570): s0 Conv3x3/2 + BatchNorm + act, s1 / s2 MBConv stages (first block of
each with stride 2), a 1x1 convolution into 5 ViTBlocks' worth of tokens... in its own words `s3_pe`, `s3`, `s4_pe`
(Conv 2x2 / 2), `s4`, then average pool and a Linear head.  There is no reference output to compare the whole model with
(parity unpinned); its blocks are the golden-pinned MBConv and ViTBlock, and the model is checked against the oracle's
restatement of the same composition.

Activations stay token-major [B, H*W, C] from the stem to the head: no NCHW <-> token conversion runs between blocks.
Differences from the snapshot's text: the stem convolution carries no bias (in front of a BatchNorm a bias cancels out of the
training forward and backward), and the transformer stages take `dropout` from the configuration (the snapshot hard-codes
0.1)."""
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..layers.attention import MHA
from ..layers.conv import MBConv
from ..layers.embeddings import ConvEmbedding
from ..layers.vit import ViTBlock


class CoAtNet(nn.Module):
	def __init__(self, config):
		super().__init__()
		dims, depths = list(config.dims), list(config.depths)
		self.chw = config.chw
		self.act_code = L.ACT_GELU if config.activation.lower() == "gelu" else L.ACT_SILU
		self.s0 = ConvEmbedding(config.chw[0], dims[0], 3, 2, 1)
		self.s1 = self._conv_stage(dims[0], dims[1], depths[0], config.activation)
		self.s2 = self._conv_stage(dims[1], dims[2], depths[1], config.activation)
		self.s3_pe = nn.Conv2d(dims[2], dims[3], 1)
		self.s3 = self._transformer_stage(dims[3], depths[2], config)
		self.s4_pe = nn.Conv2d(dims[3], dims[4], 2, 2)
		self.s4 = self._transformer_stage(dims[4], depths[3], config)
		self.head = nn.Linear(dims[-1], config.out_d)

	@staticmethod
	def _conv_stage(cin, cout, depth, activation):
		return nn.ModuleList([MBConv(cin if i == 0 else cout, cout, stride=2 if i == 0 else 1, activation=activation) for i in range(depth)])

	@staticmethod
	def _transformer_stage(dim, depth, config):
		class BlockConfig:
			attention = MHA
			d_model, n_head, d_ff = dim, config.num_heads, 4 * dim
			dropout = getattr(config, "dropout", 0.0)
			use_qk_norm = use_layer_scale = False
			norm, activation = config.norm, config.activation
		return nn.ModuleList([ViTBlock(BlockConfig()) for _ in range(depth)])

	def forward(self, x):
		_, _, h, w = x.shape
		conv = lambda n, k, s, p: (n + 2 * p - k) // s + 1
		x = self.s0(x)  # conv + BatchNorm, tokens out
		h, w = conv(h, 3, 2, 1), conv(w, 3, 2, 1)
		x = ops.ActFn.apply(x, self.act_code)
		for stage in (self.s1, self.s2):
			for blk in stage:
				x = blk(x, (h, w))
				if blk.stride == 2:
					h, w = conv(h, 3, 2, 1), conv(w, 3, 2, 1)
		x = ops.LinearFn.apply(x, self.s3_pe.weight.view(self.s3_pe.out_channels, -1), self.s3_pe.bias, L.ACT_NONE)  # a 1x1 convolution on tokens
		for blk in self.s3:
			x, _ = blk(x)
		x = ops.ConvTokensFn.apply(x, self.s4_pe.weight, self.s4_pe.bias, (2, 2, 0, (h, w)))
		for blk in self.s4:
			x, _ = blk(x)
		return ops.LinearFn.apply(ops.MeanPoolFn.apply(x), self.head.weight, self.head.bias, L.ACT_NONE)
