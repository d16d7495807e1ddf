"""Token embeddings with the reference's interface (layers/embeddings.py:7-28 there)."""
import torch
import torch.nn as nn

from .. import ops


def patchify(images, n_patch):
	"""[B, C, H, W] -> [B, n_patch^2, C*hp*wp], patch pixels flattened channel-major — the gather kernel
	vitk_patchify.  (The fused ViT model never calls this: its patch GEMM gathers in the A-tile loader.)"""
	_, _, h, w = images.shape
	assert h == w, "Patchify method is implemented for square images only"
	return ops.PatchifyFn.apply(images, n_patch)


class ConvEmbedding(nn.Module):
	"""Conv2d(bias=False) -> BatchNorm2d -> tokens [B, h*w, C_out] (CvT).  Parameters live in `.conv` and
	`.norm` exactly as in the reference; the computation is an im2col gather + MFMA GEMM + a token-major
	BatchNorm kernel, so the output is produced directly in token layout."""

	def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
		super().__init__()
		self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding, bias=False)
		self.norm = nn.BatchNorm2d(out_channels)

	def forward(self, x, hw=None):
		"""x: NCHW image/feature map, or token-major [B, h*w, C] when hw=(h, w) is given (used by the fused CvT model to
		skip the 'b (h w) c -> b c h w' round trip)."""
		return ops.conv_embedding(self, x, hw)
