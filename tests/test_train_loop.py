"""Host logic of the epoch loop (CPU): same averages as the reference's per-step `.item()` loop (train.py:25-66)."""
import torch
import torch.nn as nn

from vit_amd.train_loop import run_one_epoch


def _reference_style_epoch(model, batches, criterion, optimizer, train):
	model.train(train)
	total_loss, correct, total = 0.0, 0, 0
	with torch.set_grad_enabled(train):
		for x, y in batches:
			if train:
				optimizer.zero_grad()
			y_hat = model(x)
			loss = criterion(y_hat, y)
			if train:
				loss.backward()
				optimizer.step()
			total_loss += loss.detach().cpu().item()
			correct += torch.sum(torch.argmax(y_hat, dim=1) == y).detach().cpu().item()
			total += len(x)
	return total_loss / len(batches), correct / total


def test_epoch_loop_matches_per_step_item_loop():
	g = torch.Generator().manual_seed(0)
	batches = [(torch.randn(8, 12, generator=g), torch.randint(0, 5, (8,), generator=g)) for _ in range(6)]
	batches.append((torch.randn(3, 12, generator=g), torch.randint(0, 5, (3,), generator=g)))  # ragged last batch
	crit = nn.CrossEntropyLoss()
	for train in (True, False):
		torch.manual_seed(1)
		m1 = nn.Sequential(nn.Linear(12, 16), nn.ReLU(), nn.Linear(16, 5))
		torch.manual_seed(1)
		m2 = nn.Sequential(nn.Linear(12, 16), nn.ReLU(), nn.Linear(16, 5))
		o1 = torch.optim.SGD(m1.parameters(), lr=0.1)
		o2 = torch.optim.SGD(m2.parameters(), lr=0.1)
		ref = _reference_style_epoch(m1, batches, crit, o1, train)
		got = run_one_epoch(m2, batches, crit, torch.device("cpu"), trainer=o2, state="train" if train else "eval")
		assert abs(ref[0] - got[0]) < 1e-6 and abs(ref[1] - got[1]) < 1e-12
		for p1, p2 in zip(m1.parameters(), m2.parameters()):
			assert torch.equal(p1, p2)


def test_epoch_loop_tuple_outputs_and_empty_loader():
	class M(nn.Module):
		def __init__(self):
			super().__init__()
			self.l = nn.Linear(4, 3)

		def forward(self, x):
			return self.l(x), 0.5
	m = M()
	x, y = torch.randn(5, 4), torch.tensor([0, 1, 2, 0, 1])
	loss, acc = run_one_epoch(m, [(x, y)], nn.CrossEntropyLoss(), torch.device("cpu"), state="eval")
	assert abs(loss - (nn.CrossEntropyLoss()(m.l(x), y).item() + 0.5)) < 1e-6
	assert run_one_epoch(m, [], nn.CrossEntropyLoss(), torch.device("cpu"), state="eval") == (0.0, 0.0)
