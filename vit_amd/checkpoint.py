"""Checkpoints in the reference's file format (train.py:121-157 there), written from and read into the flat
parameter / optimizer buffers of `dp.Trainer`.

A file holds `{'epoch', 'model_state_dict', 'optimizer_state_dict', 'best_val_loss'[, 'scheduler_state_dict',
'config', 'args', 'best_model_state_dict']}`; `optimizer_state_dict` is what `torch.optim.AdamW.state_dict()` returns
for the two parameter groups of utils.py:102-121 there (decayed parameters first, then the undecayed ones, each in
`named_parameters()` order).  The fused optimizer keeps its moments in two flat float32 buffers in registration order;
the functions below translate between the two numberings, so that a run started with the reference resumes here and
the other way round.  Files are read with `weights_only=True`: nothing in them is executed (a `config` / `args` entry
holding arbitrary Python objects makes that loader refuse the file; pass `weights_only=False` yourself only for files
you wrote).
"""
import torch

from .dp import no_decay


def _group_order(flat, decay_filter=no_decay):
	"""Indices into flat.params in the order torch.optim.AdamW numbers them: decayed group, then undecayed group."""
	dec = [i for i in flat.module_order if not decay_filter(flat.names[i])]
	nod = [i for i in flat.module_order if decay_filter(flat.names[i])]
	return dec, nod


def optimizer_state_dict(opt, decay_filter=no_decay):
	"""`torch.optim.AdamW.state_dict()` of the fused optimizer `opt` (dp.FlatAdamW).  Moments are copies."""
	opt.wait_pipelined()  # (a pipelined step may still be writing the buffers)
	flat = opt.flat
	dec, nod = _group_order(flat, decay_filter)
	state = {}
	if opt.step_count > 0:
		for k, i in enumerate(dec + nod):
			o, n, shape = flat.offsets[i], flat.params[i].numel(), flat.params[i].shape
			state[k] = {
				"step": torch.tensor(float(opt.step_count)),
				"exp_avg": opt.m[o:o + n].view(shape).clone(),
				"exp_avg_sq": opt.v[o:o + n].view(shape).clone(),
			}
	common = {"lr": opt.lr, "betas": tuple(opt.betas), "eps": opt.eps, "amsgrad": False, "maximize": False, "foreach": None,
		"capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True}
	groups = [
		dict(common, weight_decay=opt.weight_decay, params=list(range(len(dec)))),
		dict(common, weight_decay=0.0, params=list(range(len(dec), len(dec) + len(nod)))),
	]
	# per-group learning rates and what schedulers keep in the groups (initial_lr ...)
	live = {0: getattr(opt, "_decay_group", None), 1: getattr(opt, "_nodecay_group", None)}
	for k, gi in live.items():
		if gi is not None:
			for key, val in opt.param_groups[gi].items():
				if key not in ("params", "betas"):
					groups[k][key] = val
	return {"state": state, "param_groups": groups}


def load_optimizer_state_dict(opt, sd, decay_filter=no_decay):
	"""Fill the fused optimizer's moments, step count and hyper-parameters from a `torch.optim.AdamW.state_dict()`
	whose groups follow the reference's rule.  Raises ValueError when group sizes or tensor shapes do not fit."""
	opt.wait_pipelined()  # (a pipelined step may still be writing the buffers)
	flat = opt.flat
	dec, nod = _group_order(flat, decay_filter)
	groups = sd["param_groups"]
	if len(groups) != 2 or len(groups[0]["params"]) != len(dec) or len(groups[1]["params"]) != len(nod):
		raise ValueError(f"optimizer state has groups of {[len(g['params']) for g in groups]} parameters, this model has [{len(dec)}, {len(nod)}]")
	order = {}
	for k, i in zip(groups[0]["params"], dec):
		order[k] = i
	for k, i in zip(groups[1]["params"], nod):
		order[k] = i
	steps = set()
	opt.m.zero_()
	opt.v.zero_()
	for k, st in sd["state"].items():
		i = order[int(k)]
		o, n, shape = flat.offsets[i], flat.params[i].numel(), flat.params[i].shape
		if tuple(st["exp_avg"].shape) != tuple(shape):
			raise ValueError(f"moment of {flat.names[i]} has shape {tuple(st['exp_avg'].shape)}, parameter has {tuple(shape)}")
		opt.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
		opt.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
		steps.add(int(float(st["step"])))
	if len(steps) > 1:
		raise ValueError(f"parameters at different step counts {sorted(steps)}: the fused optimizer keeps one count")
	opt.set_step(steps.pop() if steps else 0)
	g0 = groups[0]
	opt.lr, opt.betas, opt.eps, opt.weight_decay = g0["lr"], tuple(g0["betas"]), g0["eps"], g0["weight_decay"]
	live = {0: getattr(opt, "_decay_group", None), 1: getattr(opt, "_nodecay_group", None)}
	for k, gi in live.items():
		if gi is not None:
			for key in ("lr", "initial_lr"):
				if key in groups[k]:
					opt.param_groups[gi][key] = groups[k][key]


def load_model_state_dict(trainer_or_none, module, sd, strict=True):
	"""`module.load_state_dict(sd)` that keeps the parameters inside the trainer's flat buffer (load_state_dict copies in
	place) and then refreshes the bfloat16 / transposed copies the kernels read."""
	if trainer_or_none is not None:
		trainer_or_none.opt.wait_pipelined()
	out = module.load_state_dict(sd, strict=strict)
	if trainer_or_none is not None:
		trainer_or_none.opt.refresh_copies()
	return out


def save_checkpoint(path, module, trainer, epoch, best_val_loss=float("inf"), scheduler=None, config=None, args=None, best=False):
	"""train.py:121-145 there, field for field (including the `_best.pt` side file and the `epoch + 1` convention)."""
	trainer.opt.wait_pipelined()  # (module.state_dict() below reads the parameters)
	ck = {
		"epoch": epoch + 1,
		"model_state_dict": {k: v.detach().clone() for k, v in module.state_dict().items()},
		"optimizer_state_dict": optimizer_state_dict(trainer.opt),
		"best_val_loss": best_val_loss,
	}
	if best:
		torch.save(ck, path.replace(".pt", "_best.pt"))
		ck["best_model_state_dict"] = ck["model_state_dict"]
	if scheduler is not None:
		ck["scheduler_state_dict"] = scheduler.state_dict()
	if config:
		ck["config"] = config
	if args:
		ck["args"] = args
	torch.save(ck, path)
	return ck


def load_checkpoint(checkpoint, module, trainer=None, scheduler=None, map_location=None, weights_only=True):
	"""train.py:147-157 there.  `checkpoint` is a path or an already loaded dict.  Returns (start_epoch, best_val_loss).
	Without a trainer only the model weights are loaded (evaluation of reference-trained weights)."""
	if not isinstance(checkpoint, dict):
		checkpoint = torch.load(checkpoint, map_location=map_location, weights_only=weights_only)
	load_model_state_dict(trainer, module, checkpoint["model_state_dict"])
	if trainer is not None and "optimizer_state_dict" in checkpoint:
		load_optimizer_state_dict(trainer.opt, checkpoint["optimizer_state_dict"])
	if scheduler is not None and "scheduler_state_dict" in checkpoint:
		scheduler.load_state_dict(checkpoint["scheduler_state_dict"])
	return checkpoint.get("epoch", 0), checkpoint.get("best_val_loss", float("inf"))
