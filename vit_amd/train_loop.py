"""Epoch loop with the shape of the reference's Trainer.run_one_epoch (train.py:25-66 there) and without its per-step
host synchronisation: the reference moves the loss and the correct count to the CPU with `.item()` on every step (two
device syncs per step, which at 32x32 inputs cost more than the step itself); here both are accumulated on the device
and read once per epoch, and the batches are copied with non_blocking=True (effective with a pinned-memory loader).

The returned (avg_loss, accuracy) are defined as there: sum of per-step losses / number of steps, correct / samples.
In the training state the step is zero_grad -> forward -> loss -> backward -> optimizer step; `trainer` may be a
vit_amd.dp.Trainer (flat buffers, fused AdamW, RCCL gradient sync) or any object with zero_grad() and step()
(torch.optim optimizers qualify).  A captured step (vit_amd.dp.GraphedStep) can be passed as `graphed_step`; it then
replaces forward/backward/update and returns the loss tensor, or (loss, logits) when its loss_fn returns both; without
logits the epoch's accuracy is reported as NaN (unknown), not 0.
"""
import torch


def compute_loss(criterion, y_hat, y):
	"""Trainer.compute_loss (train.py:68-74): models may return (logits, regularization_loss)."""
	if isinstance(y_hat, tuple):
		logits, reg = y_hat
		return criterion(logits, y) + reg
	return criterion(y_hat, y)


def run_one_epoch(model, dataloader, criterion, device, trainer=None, state="train", graphed_step=None):
	is_training = state == "train"
	model.train(is_training)
	if is_training and trainer is None and graphed_step is None:
		raise ValueError("training needs a trainer / optimizer (zero_grad, step) or a graphed step")
	total_loss = torch.zeros((), dtype=torch.float32, device=device)
	correct = torch.zeros((), dtype=torch.int64, device=device)
	total, steps, counted = 0, 0, 0
	with torch.set_grad_enabled(is_training):
		for x, y in dataloader:
			x = x.to(device, non_blocking=True)
			y = y.to(device, non_blocking=True)
			if is_training and graphed_step is not None:
				out = graphed_step(x, y)
				loss, logits = out if isinstance(out, tuple) else (out, None)
			else:
				if is_training:
					trainer.zero_grad()
				y_hat = model(x)
				loss = compute_loss(criterion, y_hat, y)
				if is_training:
					loss.backward()
					trainer.step()
				logits = y_hat[0] if isinstance(y_hat, tuple) else y_hat
			total_loss += loss.detach().float()
			if logits is not None:
				correct += (logits.detach().argmax(dim=1) == y).sum()
				counted += len(x)
			total += len(x)
			steps += 1
	if steps == 0:
		return 0.0, 0.0
	# the only device -> host reads of the epoch
	return total_loss.item() / steps, (correct.item() / counted if counted == total else float("nan"))
