"""FeedForward with the reference's interface (layers/transformers.py:5-57 there): fc_in -> activation
(or gate * activation for swiglu/geglu) -> dropout -> fc_out -> dropout, as two MFMA GEMMs with bias,
activation and the activation derivative fused into their epilogues."""
import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from .dropout import Dropout

SUPPORTED = ("swiglu", "silu", "geglu", "gelu", "reglu", "relu", "reglu6", "relu6")


class FeedForward(nn.Module):
	def __init__(self, config):
		super().__init__()
		self.activation = config.activation.lower()
		if self.activation not in SUPPORTED:
			raise ValueError(f"Unknown activation type: {self.activation}")
		self.uses_gate = self.activation in ("swiglu", "geglu", "reglu", "reglu6")
		# only swiglu / geglu double fc_in and gate; reglu / reglu6 act as plain relu / relu6
		self.gated = self.activation in ("swiglu", "geglu")
		self.act_code = {"swiglu": L.ACT_SILU, "silu": L.ACT_SILU, "geglu": L.ACT_GELU, "gelu": L.ACT_GELU,
			"reglu6": L.ACT_RELU6, "relu6": L.ACT_RELU6, "reglu": L.ACT_RELU, "relu": L.ACT_RELU}[self.activation]
		self.fc_in = nn.Linear(config.d_model, config.d_ff * (2 if self.gated else 1))
		self.fc_out = nn.Linear(config.d_ff, config.d_model)
		self.dropout1 = Dropout(config.dropout)
		self.dropout2 = Dropout(config.dropout)

	def forward(self, x: torch.Tensor) -> torch.Tensor:
		if self.dropout1.active():
			# dropout between the two GEMMs: run them as separate nodes
			h = ops.LinearFn.apply(x, self.fc_in.weight, self.fc_in.bias, L.ACT_NONE if self.gated else self.act_code)
			if self.gated:
				h = ops.GatedActFn.apply(h, self.act_code)
			h = self.dropout1(h)
			y = ops.LinearFn.apply(h, self.fc_out.weight, self.fc_out.bias, L.ACT_NONE)
		else:
			y = ops.FeedForwardFn.apply(x, self.fc_in.weight, self.fc_in.bias, self.fc_out.weight, self.fc_out.bias, self.act_code, self.gated)
		return self.dropout2(y)
