"""Placeholder for the reference's layers/conv.py (MBConv, used only by its CoAtNet, which cannot be
constructed).  Not on the MI355X hot path."""
import torch.nn as nn


class MBConv(nn.Module):
	def __init__(self, *args, **kwargs):
		super().__init__()
		raise NotImplementedError("MBConv is outside the MI355X hot path of this build")
