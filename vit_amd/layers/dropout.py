"""nn.Dropout stand-in that runs on the vitk dropout kernel (counter-based RNG, mask regenerated in
the backward).  Inactive (eval mode or p == 0) it returns its input untouched."""
import torch.nn as nn

from .. import ops


class Dropout(nn.Dropout):
	def active(self):
		return self.training and self.p > 0.0

	def forward(self, x):
		if not self.active():
			return x
		return ops.dropout(x, self.p)
