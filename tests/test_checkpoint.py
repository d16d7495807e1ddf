"""CPU: checkpoints in the reference's format (train.py:121-157 there).  The optimizer state is checked against a real
torch.optim.AdamW built with the reference's parameter groups (utils.py:102-121 there), in both directions."""
import os

import pytest
import torch
import torch.nn as nn

from vit_amd import checkpoint as ck
from vit_amd.dp import FlatAdamW, FlatModel, no_decay


class Small(nn.Module):
	def __init__(self):
		super().__init__()
		self.fc_in = nn.Linear(6, 10)
		self.norm1 = nn.LayerNorm(10)
		self.attn_layer_scale = nn.Parameter(torch.full((10,), 1e-4))
		self.fc_out = nn.Linear(10, 3, bias=False)

	def forward(self, x):
		return self.fc_out(self.norm1(self.fc_in(x)) * self.attn_layer_scale)


def reference_groups(model, wd):
	"""The grouping rule restated from utils.py:102-121 there."""
	dec, nod = [], []
	for n, p in model.named_parameters():
		(nod if (n.endswith("bias") or "norm" in n.lower() or "layerscale" in n.lower()) else dec).append(p)
	return [{"params": dec, "weight_decay": wd}, {"params": nod, "weight_decay": 0.0}]


class FakeTrainer:
	def __init__(self, module):
		self.flat = FlatModel(module)
		self.opt = FlatAdamW(self.flat, lr=3e-3, weight_decay=0.05, bf16_shadow=False)
		self.opt.refresh_copies = lambda: None  # no device copies on the CPU


def test_no_decay_rule_matches_reference_rule():
	m = Small()
	names = [n for n, _ in m.named_parameters()]
	assert [no_decay(n) for n in names] == [n.endswith("bias") or "norm" in n.lower() or "layerscale" in n.lower() for n in names]
	assert not no_decay("blocks.0.attn_layer_scale")  # decayed, as in the reference
	assert no_decay("norm1.weight") and no_decay("fc_in.bias") and not no_decay("bias_table.weight")


def test_torch_adamw_state_loads_into_flat_buffers_and_back(tmp_path):
	torch.manual_seed(0)
	ref = Small()
	opt = torch.optim.AdamW(reference_groups(ref, 0.05), lr=3e-3)
	for _ in range(3):
		opt.zero_grad()
		ref(torch.randn(4, 6)).square().sum().backward()
		opt.step()
	path = str(tmp_path / "ref.pt")
	torch.save({"epoch": 7, "model_state_dict": ref.state_dict(), "optimizer_state_dict": opt.state_dict(), "best_val_loss": 0.25}, path)

	ours = Small()
	tr = FakeTrainer(ours)
	epoch, best = ck.load_checkpoint(path, ours, tr)
	assert (epoch, best) == (7, 0.25)
	assert tr.opt.step_count == 3 and tr.opt.lr == 3e-3 and tr.opt.weight_decay == 0.05
	for (n, p), q in zip(ours.named_parameters(), ref.parameters()):
		assert torch.equal(p, q), n
		assert p.data_ptr() >= tr.flat.param.data_ptr() and p.data_ptr() < tr.flat.param.data_ptr() + 4 * tr.flat.numel, "parameter left the flat buffer"
	byname = dict(ref.named_parameters())
	for n, o, p in zip(tr.flat.names, tr.flat.offsets, tr.flat.params):
		st = opt.state[byname[n]]
		assert torch.equal(tr.opt.m[o:o + p.numel()].view(p.shape), st["exp_avg"]), n
		assert torch.equal(tr.opt.v[o:o + p.numel()].view(p.shape), st["exp_avg_sq"]), n

	# and back: our state_dict loads into a fresh torch AdamW and equals the original
	out = str(tmp_path / "ours.pt")
	ck.save_checkpoint(out, ours, tr, epoch=6, best_val_loss=0.25, best=True)
	assert os.path.exists(out.replace(".pt", "_best.pt"))
	loaded = torch.load(out, weights_only=True)
	assert loaded["epoch"] == 7 and "best_model_state_dict" in loaded
	back = Small()
	back.load_state_dict(loaded["model_state_dict"])
	opt2 = torch.optim.AdamW(reference_groups(back, 0.05), lr=1.0)
	opt2.load_state_dict(loaded["optimizer_state_dict"])
	assert opt2.param_groups[0]["lr"] == 3e-3 and opt2.param_groups[1]["weight_decay"] == 0.0
	for (n, p), (_, q) in zip(back.named_parameters(), ref.named_parameters()):
		assert torch.equal(opt2.state[p]["exp_avg"], opt.state[q]["exp_avg"]), n
		assert torch.equal(opt2.state[p]["exp_avg_sq"], opt.state[q]["exp_avg_sq"]), n
		assert float(opt2.state[p]["step"]) == 3.0
	# the restored torch optimizer takes the same next step as the original
	x = torch.randn(4, 6)
	for m_, o_ in ((ref, opt), (back, opt2)):
		o_.zero_grad()
		m_(x).square().sum().backward()
		o_.step()
	for p, q in zip(back.parameters(), ref.parameters()):
		assert torch.equal(p, q)


def test_fresh_optimizer_has_empty_state_and_mismatches_raise():
	tr = FakeTrainer(Small())
	sd = ck.optimizer_state_dict(tr.opt)
	assert sd["state"] == {} and [len(g["params"]) for g in sd["param_groups"]] == [3, 3]
	torch.optim.AdamW(reference_groups(Small(), 0.05)).load_state_dict(sd)
	bad = {"state": {}, "param_groups": [dict(sd["param_groups"][0], params=[0]), dict(sd["param_groups"][1], params=[1, 2, 3, 4, 5])]}
	with pytest.raises(ValueError, match="groups"):
		ck.load_optimizer_state_dict(tr.opt, bad)
