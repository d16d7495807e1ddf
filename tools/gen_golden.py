"""Generate golden vectors by running the reference's own PyTorch modules on the CPU.

Run in the authoring container only (the reference checkout does not travel):

    python tools/gen_golden.py --reference /root/reference --out tests/golden

Each fixture is an .npz holding seeded inputs, the full state_dict ("sd/<key>"),
the reference's outputs, loss, every parameter gradient ("grad/<key>") and the
input gradient, plus a JSON "meta" record (config, torch/einops versions).
Nothing from the reference (source, bytecode, pickled classes) is written out:
only tensors.
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def np_(t):
	return t.detach().cpu().float().clone().numpy() if t.is_floating_point() else t.detach().cpu().clone().numpy()


def save(out, name, arrays, meta):
	import einops
	meta = dict(meta, torch=torch.__version__, einops=einops.__version__, generator="tools/gen_golden.py")
	arrays = dict(arrays)
	arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
	path = os.path.join(out, name + ".npz")
	np.savez(path, **arrays)
	print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB")


def model_fixture(model, x, y, criterion, extra=None):
	model.zero_grad()
	x = x.clone().requires_grad_(True)
	out = model(x)
	logits = out[0] if isinstance(out, tuple) else out
	loss = criterion(logits, y)
	loss.backward()
	arrays = {"x": np_(x), "y": np_(y), "logits": np_(logits), "loss": np_(loss), "grad_x": np_(x.grad)}
	for k, v in model.state_dict().items():
		arrays["sd/" + k] = np_(v)
	for k, p in model.named_parameters():
		if p.grad is not None:
			arrays["grad/" + k] = np_(p.grad)
	if extra:
		arrays.update(extra)
	return arrays


MBCONV_CASES = {"res_silu": (16, 16, 1, 4, "silu", 8), "down_gelu": (16, 32, 2, 4, "gelu", 7), "wide_relu": (8, 24, 1, 2, "relu", 6)}


def mbconv_fixture(args):
	"""MBConv (layers/conv.py:5-44): 1x1 expand -> BN -> act -> depthwise 3x3 -> BN -> act -> 1x1 project -> BN
	(+ input when stride 1 and equal channels), train and eval."""
	from layers.conv import MBConv
	g = torch.Generator().manual_seed(11)
	arrays = {}
	for nm, (ci, co, stride, expand, act, hw) in MBCONV_CASES.items():
		for training in (True, False):
			torch.manual_seed(5)
			m = MBConv(ci, co, stride=stride, expand_ratio=expand, activation=act)
			with torch.no_grad():
				for mod in m.modules():
					if isinstance(mod, nn.BatchNorm2d):
						c = mod.num_features
						mod.running_mean.copy_(0.1 * torch.randn(c, generator=g))
						mod.running_var.copy_(0.5 + torch.rand(c, generator=g))
						mod.weight.add_(0.2 * torch.randn(c, generator=g))
						mod.bias.add_(0.2 * torch.randn(c, generator=g))
			m.train(training)
			tag = f"{nm}/{'train' if training else 'eval'}"
			for kk, v in m.state_dict().items():
				arrays[f"{tag}/sd/{kk}"] = np_(v)
			x = torch.randn(3, ci, hw, hw, generator=g).requires_grad_(True)
			yv = m(x)
			dy = torch.randn(yv.shape, generator=g)
			yv.backward(dy)
			arrays[f"{tag}/x"] = np_(x); arrays[f"{tag}/y"] = np_(yv); arrays[f"{tag}/dy"] = np_(dy); arrays[f"{tag}/dx"] = np_(x.grad)
			for kk, pp in m.named_parameters():
				arrays[f"{tag}/grad/{kk}"] = np_(pp.grad)
			for kk, v in m.state_dict().items():
				if "running_" in kk:
					arrays[f"{tag}/sd_after/{kk}"] = np_(v)
	save(args.out, "mbconv", arrays, {"kind": "mbconv", "cases": {k: list(v) for k, v in MBCONV_CASES.items()}})


def teacher_fixture(args):
	"""The distillation teacher (models/resnet.py:55-85 there) with the weights of the reference's own
	checkpoints/resnet20_cifar10.pth (a bare state_dict, read with weights_only=True): state_dict, a seeded input batch and the
	eval-mode logits."""
	from models.resnet import resnet_cifar
	sd = torch.load(os.path.join(args.reference, "checkpoints", "resnet20_cifar10.pth"), map_location="cpu", weights_only=True)
	m = resnet_cifar(20, num_classes=10)
	m.load_state_dict(sd, strict=True)
	m.eval()
	g = torch.Generator().manual_seed(2020)
	x = torch.randn(8, 3, 32, 32, generator=g)
	with torch.no_grad():
		logits = m(x)
	arrays = {"x": np_(x), "logits": np_(logits)}
	for k, v in m.state_dict().items():
		arrays["sd/" + k] = np_(v)
	save(args.out, "resnet20_teacher", arrays, {"kind": "teacher", "depth": 20, "num_classes": 10, "weights": "checkpoints/resnet20_cifar10.pth"})


def ref_bf16_errors(args):
	"""What the reference itself loses when it runs in bfloat16: its ViT models (the four golden configurations) forward +
	backward under torch.autocast("cpu", bfloat16) against its own float32 run on the same seeded input, per tensor, in the
	error measure the parity tests use (max|a - b| / max|b|).  Written as JSON (numbers only) next to the goldens."""
	from config import ViTConfig
	from layers.attention import MHA
	from layers.norm import RMSNorm
	from models.vit import ViT
	ce = nn.CrossEntropyLoss()
	out = {}
	rel = lambda a, b: float(((a.detach().double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)))
	for name in ("vit_ln_gelu", "vit_p4_n65", "vit_rms_swiglu_ls", "vit_p16_geglu"):
		z = np.load(os.path.join(args.out, name + ".npz"), allow_pickle=False)
		c = json.loads(bytes(z["meta"]).decode())["config"]

		class Cfg(ViTConfig):
			pass
		for k, v in c.items():
			if k not in ("norm", "batch"):
				setattr(Cfg, k, tuple(v) if isinstance(v, list) else v)
		Cfg.attention = MHA
		Cfg.norm = nn.LayerNorm if c["norm"] == "layernorm" else RMSNorm
		Cfg.dropout = 0.0
		Cfg.use_qk_norm = False
		model = ViT(Cfg())
		model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}, strict=True)
		x = torch.from_numpy(z["x"]).requires_grad_(True)
		y = torch.from_numpy(z["y"])
		model.zero_grad()
		with torch.autocast("cpu", dtype=torch.bfloat16):
			o = model(x)
			logits = o[0] if isinstance(o, tuple) else o
			loss = ce(logits.float(), y)
		loss.backward()
		rel = lambda a, b: float(((a.detach().double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)))
		rec = {"logits": rel(logits, torch.from_numpy(z["logits"])), "grad_x": rel(x.grad, torch.from_numpy(z["grad_x"]))}
		worst = ("", 0.0)
		for k, p_ in model.named_parameters():
			e = rel(p_.grad, torch.from_numpy(z["grad/" + k]))
			rec["grad/" + k] = e
			if e > worst[1]:
				worst = (k, e)
		rec["worst_param_grad"] = {"name": worst[0], "err": worst[1]}
		out[name] = rec
		print(name, "logits", f"{rec['logits']:.3g}", "worst grad", worst)
	# the modules with BatchNorm in them: CvT (three conv embeddings), ConvEmbedding alone, MBConv (three BatchNorms)
	from config import CvTConfig
	from layers.conv import MBConv
	from layers.embeddings import ConvEmbedding
	from models.cvt import CvT
	for mode in ("train", "eval"):
		z = np.load(os.path.join(args.out, f"cvt_small_{mode}.npz"), allow_pickle=False)
		c = json.loads(bytes(z["meta"]).decode())["config"]

		class CCfg(CvTConfig):
			pass
		for k, v in c.items():
			if k not in ("norm", "batch"):
				setattr(CCfg, k, tuple(v) if isinstance(v, list) else v)
		CCfg.norm = RMSNorm
		CCfg.dropout = 0.0
		model = CvT(CCfg())
		model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}, strict=True)
		model.train(mode == "train")
		x = torch.from_numpy(z["x"]).requires_grad_(True)
		y = torch.from_numpy(z["y"])
		with torch.autocast("cpu", dtype=torch.bfloat16):
			logits = model(x)
			loss = ce(logits.float(), y)
		loss.backward()
		rec = {"logits": rel(logits, torch.from_numpy(z["logits"])), "grad_x": rel(x.grad, torch.from_numpy(z["grad_x"]))}
		worst = ("", 0.0)
		gmax = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad/"))
		for k, p_ in model.named_parameters():
			if float(np.abs(z["grad/" + k]).max()) < 1e-6 * gmax:
				continue  # an exactly-zero gradient in float32 (stage 3 attends to ONE key: its query projection receives none)
			e = rel(p_.grad, torch.from_numpy(z["grad/" + k]))
			rec["grad/" + k] = e
			if e > worst[1]:
				worst = (k, e)
		rec["worst_param_grad"] = {"name": worst[0], "err": worst[1]}
		out[f"cvt_small_{mode}"] = rec
		print(f"cvt_small_{mode}", "logits", f"{rec['logits']:.3g}", "grad_x", f"{rec['grad_x']:.3g}", "worst grad", worst)

	def leaf_module(tag, m, z, prefix):
		m.load_state_dict({k[len(prefix) + 4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix + "/sd/")}, strict=True)
		x = torch.from_numpy(z[prefix + "/x"]).requires_grad_(True)
		with torch.autocast("cpu", dtype=torch.bfloat16):
			yv = m(x)
		yv.backward(torch.from_numpy(z[prefix + "/dy"]).to(yv.dtype))
		rec = {"y": rel(yv, torch.from_numpy(z[prefix + "/y"])), "dx": rel(x.grad, torch.from_numpy(z[prefix + "/dx"]))}
		worst = ("", 0.0)
		for k, p_ in m.named_parameters():
			e = rel(p_.grad, torch.from_numpy(z[f"{prefix}/grad/{k}"]))
			rec["grad/" + k] = e
			if e > worst[1]:
				worst = (k, e)
		rec["worst_param_grad"] = {"name": worst[0], "err": worst[1]}
		out[tag] = rec
		print(tag, "y", f"{rec['y']:.3g}", "dx", f"{rec['dx']:.3g}", "worst grad", worst)
	z = np.load(os.path.join(args.out, "conv_embedding.npz"), allow_pickle=False)
	for nm, (ci, co, k, s_, p, hw) in json.loads(bytes(z["meta"]).decode())["cases"].items():
		for mode in ("train", "eval"):
			m = ConvEmbedding(ci, co, k, s_, p)
			m.train(mode == "train")
			leaf_module(f"conv_embedding/{nm}/{mode}", m, z, f"{nm}/{mode}")
	z = np.load(os.path.join(args.out, "mbconv.npz"), allow_pickle=False)
	for nm, (ci, co, stride, expand, act, hw) in json.loads(bytes(z["meta"]).decode())["cases"].items():
		for mode in ("train", "eval"):
			m = MBConv(ci, co, stride=stride, expand_ratio=expand, activation=act)
			m.train(mode == "train")
			leaf_module(f"mbconv/{nm}/{mode}", m, z, f"{nm}/{mode}")
	with open(os.path.join(args.out, "ref_bf16_errors.json"), "w") as f:
		json.dump({"what": "reference modules, torch.autocast('cpu', bfloat16) vs their float32 run; max|a-b|/max|b| per tensor", "torch": torch.__version__, "errors": out}, f, indent=1, sort_keys=True)


def dyt_fixture(args):
	"""DyT (layers/norm.py:35-45 there) alone, and a ViTBlock built with config.norm = DyT (layers/vit.py:9-60): inputs,
	parameters, outputs, every gradient, float32 on the CPU."""
	from config import ViTConfig
	from layers.norm import DyT
	from layers.vit import ViTBlock
	g = torch.Generator().manual_seed(77)
	arrays = {}
	m = DyT(40, init_alpha=0.7)
	with torch.no_grad():
		m.gamma.add_(0.3 * torch.randn(40, generator=g))
		m.beta.add_(0.3 * torch.randn(40, generator=g))
	x = (2.0 * torch.randn(3, 7, 40, generator=g)).requires_grad_(True)
	y = m(x)
	dy = torch.randn(y.shape, generator=g)
	y.backward(dy)
	arrays.update({"mod/x": np_(x), "mod/y": np_(y), "mod/dy": np_(dy), "mod/dx": np_(x.grad)})
	for k, p_ in m.named_parameters():
		arrays["mod/sd/" + k] = np_(p_)
		arrays["mod/grad/" + k] = np_(p_.grad)
	torch.manual_seed(9)
	from layers.attention import MHA

	class Cfg(ViTConfig):
		d_model = 48; n_head = 4; d_ff = 96; norm = DyT; attention = MHA; activation = "gelu"; dropout = 0.0; use_layer_scale = False; use_qk_norm = False
	blk = ViTBlock(Cfg())
	with torch.no_grad():
		for k, p_ in blk.named_parameters():
			if "norm" in k:
				p_.add_(0.2 * torch.randn(p_.shape, generator=g))
	x = torch.randn(2, 9, 48, generator=g).requires_grad_(True)
	out = blk(x)
	yb = out[0] if isinstance(out, tuple) else out
	dy = torch.randn(yb.shape, generator=g)
	yb.backward(dy)
	arrays.update({"blk/x": np_(x), "blk/y": np_(yb), "blk/dy": np_(dy), "blk/dx": np_(x.grad)})
	for k, v in blk.state_dict().items():
		arrays["blk/sd/" + k] = np_(v)
	for k, p_ in blk.named_parameters():
		arrays["blk/grad/" + k] = np_(p_.grad)
	save(args.out, "dyt", arrays, {"kind": "dyt", "block": {"d_model": 48, "n_head": 4, "d_ff": 96, "activation": "gelu", "use_layer_scale": False}})


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--reference", default="/root/reference")
	ap.add_argument("--out", default="tests/golden")
	ap.add_argument("--only", default="", help="generate only this fixture (mbconv, teacher, dyt, ref_bf16)")
	args = ap.parse_args()
	sys.path.insert(0, args.reference)
	os.makedirs(args.out, exist_ok=True)
	torch.set_num_threads(4)

	if args.only == "mbconv":
		mbconv_fixture(args)
		return
	if args.only == "teacher":
		teacher_fixture(args)
		return
	if args.only == "dyt":
		dyt_fixture(args)
		return
	if args.only == "ref_bf16":
		ref_bf16_errors(args)
		return
	from config import ViTConfig, CvTConfig
	from layers.attention import MHA, ConvAttention
	from layers.embeddings import ConvEmbedding, patchify
	from layers.norm import RMSNorm
	from layers.transformers import FeedForward
	from layers.vit import ViTBlock
	from models.vit import ViT
	from models.cvt import CvT

	ce = nn.CrossEntropyLoss()

	# ---- ViT variants (models/vit.py) ---------------------------------------------------
	vit_cfgs = {
		# LayerNorm + GELU, dh=64, 17 tokens (8x8-pixel patches)
		"vit_ln_gelu": dict(d_model=128, n_head=2, n_block=2, n_patch=4, d_ff=256, activation="gelu", norm="layernorm",
			use_layer_scale=False, out_d=10, chw=(3, 32, 32), batch=3),
		# ViT-Tiny patch-4 geometry: 65 tokens, 4x4-pixel patches (K=48)
		"vit_p4_n65": dict(d_model=128, n_head=2, n_block=1, n_patch=8, d_ff=128, activation="gelu", norm="layernorm",
			use_layer_scale=False, out_d=10, chw=(3, 32, 32), batch=2),
		# the shipped default's flavour with MHA: dh=12, swiglu, RMSNorm, LayerScale
		"vit_rms_swiglu_ls": dict(d_model=48, n_head=4, n_block=2, n_patch=4, d_ff=96, activation="swiglu", norm="rmsnorm",
			use_layer_scale=True, out_d=100, chw=(3, 32, 32), batch=2),
		# 16x16 patches as in ViT-B/16 (K=768), 5 tokens, geglu
		"vit_p16_geglu": dict(d_model=64, n_head=1, n_block=1, n_patch=2, d_ff=64, activation="geglu", norm="layernorm",
			use_layer_scale=True, out_d=7, chw=(3, 32, 32), batch=2),
	}
	for name, c in vit_cfgs.items():
		class Cfg(ViTConfig):
			pass
		for k, v in c.items():
			if k not in ("norm", "batch"):
				setattr(Cfg, k, v)
		Cfg.attention = MHA
		Cfg.norm = nn.LayerNorm if c["norm"] == "layernorm" else RMSNorm
		Cfg.dropout = 0.0
		Cfg.use_qk_norm = False
		torch.manual_seed(0)
		model = ViT(Cfg())
		if c["norm"] == "rmsnorm":
			# RMSNorm weights are zero-initialised (layers/norm.py:17); perturb so the (1+w) path is exercised
			with torch.no_grad():
				for n_, p in model.named_parameters():
					if n_.endswith("norm.weight"):
						p.add_(0.1 * torch.randn_like(p))
		if c["use_layer_scale"]:
			with torch.no_grad():
				for n_, p in model.named_parameters():
					if n_.endswith("layer_scale"):
						p.copy_(0.5 + 0.5 * torch.rand_like(p))
		g = torch.Generator().manual_seed(1234)
		x = torch.randn(c["batch"], *c["chw"], generator=g)
		y = torch.randint(0, c["out_d"], (c["batch"],), generator=g)
		arrays = model_fixture(model, x, y, ce)
		arrays["patches"] = np_(patchify(x, c["n_patch"]))
		save(args.out, name, arrays, {"kind": "vit", "config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in c.items()}})

	# ---- CvT (models/cvt.py) --------------------------------------------------------------
	cvt_c = dict(chw=(3, 32, 32), out_d=10, s1_emb_dim=32, s1_heads=1, s1_depth=1, s1_mlp_ratio=2.0,
		s2_emb_dim=64, s2_heads=2, s2_depth=1, s2_mlp_ratio=2.0, s3_emb_dim=128, s3_heads=2, s3_depth=1, s3_mlp_ratio=1.0,
		norm="rmsnorm", activation="silu", use_layer_scale=False, batch=2)
	class CCfg(CvTConfig):
		pass
	for k, v in cvt_c.items():
		if k not in ("norm", "batch"):
			setattr(CCfg, k, v)
	CCfg.norm = RMSNorm
	CCfg.dropout = 0.0
	for training in (True, False):
		torch.manual_seed(0)
		model = CvT(CCfg())
		with torch.no_grad():
			for n_, p in model.named_parameters():
				if n_.endswith("norm.weight") and p.dim() == 1 and "embedding" not in n_:
					p.add_(0.1 * torch.randn_like(p))
			for n_, b_ in model.named_buffers():
				if n_.endswith("running_mean"):
					b_.copy_(0.1 * torch.randn_like(b_))
				if n_.endswith("running_var"):
					b_.copy_(0.5 + torch.rand_like(b_))
		model.train(training)
		g = torch.Generator().manual_seed(1234)
		x = torch.randn(cvt_c["batch"], 3, 32, 32, generator=g)
		y = torch.randint(0, 10, (cvt_c["batch"],), generator=g)
		sd_before = {k: v.clone() for k, v in model.state_dict().items()}
		arrays = model_fixture(model, x, y, ce)
		# state_dict as it was BEFORE the forward (BatchNorm running stats are updated by a training forward)
		for k, v in sd_before.items():
			arrays["sd/" + k] = np_(v)
		for k, v in model.state_dict().items():
			if "running_" in k:
				arrays["sd_after/" + k] = np_(v)
		cfgj = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cvt_c.items()}
		for s in (1, 2, 3):
			for f in ("emb_kernel", "emb_stride", "emb_pad", "qkv_kernel", "qkv_stride", "qkv_pad"):
				cfgj[f"s{s}_{f}"] = getattr(CCfg, f"s{s}_{f}")
		save(args.out, "cvt_small_" + ("train" if training else "eval"), arrays, {"kind": "cvt", "training": training, "config": cfgj})

	# ---- leaf modules -----------------------------------------------------------------------
	g = torch.Generator().manual_seed(7)
	# FeedForward with every supported activation (layers/transformers.py:14-57)
	arrays = {}
	for act in ("swiglu", "silu", "geglu", "gelu", "reglu", "relu", "reglu6", "relu6"):
		class FCfg:
			d_model = 32; d_ff = 64; dropout = 0.0; activation = act
		torch.manual_seed(1)
		ff = FeedForward(FCfg())
		x = (2.5 * torch.randn(2, 5, 32, generator=g)).requires_grad_(True)
		yv = ff(x)
		dy = torch.randn(yv.shape, generator=g)
		yv.backward(dy)
		arrays[f"{act}/x"] = np_(x); arrays[f"{act}/y"] = np_(yv); arrays[f"{act}/dy"] = np_(dy); arrays[f"{act}/dx"] = np_(x.grad)
		for k, p in ff.named_parameters():
			arrays[f"{act}/sd/{k}"] = np_(p); arrays[f"{act}/grad/{k}"] = np_(p.grad)
	save(args.out, "feedforward", arrays, {"kind": "feedforward", "d_model": 32, "d_ff": 64})

	# RMSNorm (layers/norm.py:7-31) and nn.LayerNorm
	arrays = {}
	for nm, mod in (("rmsnorm", RMSNorm(40)), ("rmsnorm_nooffset", RMSNorm(40, add_unit_offset=False)), ("layernorm", nn.LayerNorm(40))):
		with torch.no_grad():
			for p in mod.parameters():
				p.add_(0.3 * torch.randn(p.shape, generator=g))
		x = (1.5 * torch.randn(3, 7, 40, generator=g) + 0.5).requires_grad_(True)
		yv = mod(x)
		dy = torch.randn(yv.shape, generator=g)
		yv.backward(dy)
		arrays[f"{nm}/x"] = np_(x); arrays[f"{nm}/y"] = np_(yv); arrays[f"{nm}/dy"] = np_(dy); arrays[f"{nm}/dx"] = np_(x.grad)
		for k, p in mod.named_parameters():
			arrays[f"{nm}/sd/{k}"] = np_(p); arrays[f"{nm}/grad/{k}"] = np_(p.grad)
	save(args.out, "norms", arrays, {"kind": "norms", "dim": 40})

	# MHA (layers/attention.py:8-48): dh in {64, 12, 24}
	arrays = {}
	for nm, (d, h, s) in {"dh64": (128, 2, 17), "dh12": (48, 4, 17), "dh24": (48, 2, 9), "dh64_n65": (64, 1, 65)}.items():
		class MCfg:
			d_model = d; n_head = h; dropout = 0.0; use_qk_norm = False
		torch.manual_seed(2)
		m = MHA(MCfg())
		x = torch.randn(2, s, d, generator=g).requires_grad_(True)
		yv = m(x)
		dy = torch.randn(yv.shape, generator=g)
		yv.backward(dy)
		arrays[f"{nm}/x"] = np_(x); arrays[f"{nm}/y"] = np_(yv); arrays[f"{nm}/dy"] = np_(dy); arrays[f"{nm}/dx"] = np_(x.grad)
		for k, p in m.named_parameters():
			arrays[f"{nm}/sd/{k}"] = np_(p); arrays[f"{nm}/grad/{k}"] = np_(p.grad)
	save(args.out, "mha", arrays, {"kind": "mha", "cases": {"dh64": [128, 2, 17], "dh12": [48, 4, 17], "dh24": [48, 2, 9], "dh64_n65": [64, 1, 65]}})

	# ConvAttention (layers/attention.py:50-93): kv stride 1 and 2
	arrays = {}
	for nm, (dim, h, hw, stride) in {"s1": (64, 1, 8, 1), "s2": (64, 2, 8, 2), "s2_odd": (32, 1, 5, 2)}.items():
		torch.manual_seed(3)
		m = ConvAttention(dim, h, kernel_size=3, qkv_stride=stride, padding=1, dropout=0.0)
		x = torch.randn(2, hw * hw, dim, generator=g).requires_grad_(True)
		yv = m(x, hw, hw)
		dy = torch.randn(yv.shape, generator=g)
		yv.backward(dy)
		arrays[f"{nm}/x"] = np_(x); arrays[f"{nm}/y"] = np_(yv); arrays[f"{nm}/dy"] = np_(dy); arrays[f"{nm}/dx"] = np_(x.grad)
		for k, p in m.named_parameters():
			arrays[f"{nm}/sd/{k}"] = np_(p); arrays[f"{nm}/grad/{k}"] = np_(p.grad)
	save(args.out, "conv_attention", arrays, {"kind": "conv_attention", "cases": {"s1": [64, 1, 8, 1], "s2": [64, 2, 8, 2], "s2_odd": [32, 1, 5, 2]}})

	# ConvEmbedding (layers/embeddings.py:18-28): k7s4p2 and k3s2p1, train and eval
	arrays = {}
	for nm, (ci, co, k, s, p, hw) in {"k7s4p2": (3, 32, 7, 4, 2, 32), "k3s2p1": (32, 64, 3, 2, 1, 8)}.items():
		for training in (True, False):
			torch.manual_seed(4)
			m = ConvEmbedding(ci, co, k, s, p)
			with torch.no_grad():
				m.norm.running_mean.copy_(0.1 * torch.randn(co, generator=g))
				m.norm.running_var.copy_(0.5 + torch.rand(co, generator=g))
				m.norm.weight.add_(0.2 * torch.randn(co, generator=g))
				m.norm.bias.add_(0.2 * torch.randn(co, generator=g))
			m.train(training)
			tag = f"{nm}/{'train' if training else 'eval'}"
			for kk, v in m.state_dict().items():
				arrays[f"{tag}/sd/{kk}"] = np_(v)
			x = torch.randn(2, ci, hw, hw, generator=g).requires_grad_(True)
			yv = m(x)
			dy = torch.randn(yv.shape, generator=g)
			yv.backward(dy)
			arrays[f"{tag}/x"] = np_(x); arrays[f"{tag}/y"] = np_(yv); arrays[f"{tag}/dy"] = np_(dy); arrays[f"{tag}/dx"] = np_(x.grad)
			for kk, pp in m.named_parameters():
				arrays[f"{tag}/grad/{kk}"] = np_(pp.grad)
			for kk, v in m.state_dict().items():
				if "running_" in kk:
					arrays[f"{tag}/sd_after/{kk}"] = np_(v)
	save(args.out, "conv_embedding", arrays, {"kind": "conv_embedding", "cases": {"k7s4p2": [3, 32, 7, 4, 2, 32], "k3s2p1": [32, 64, 3, 2, 1, 8]}})

	# DeiT loss (deit_trainer.py:23-52).  `train.py` imports tensorboard, which is not installed:
	# an empty stand-in module for torch.utils.tensorboard lets the reference trainer classes import.
	tb = types.ModuleType("torch.utils.tensorboard")
	tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None, "add_scalar": lambda self, *a, **k: None})
	sys.modules["torch.utils.tensorboard"] = tb
	from deit_trainer import DeiTTrainer
	arrays = {}
	cases = {"a05_t1": (0.5, 1.0, 0.0), "a03_t3_ls01": (0.3, 3.0, 0.1), "a1_t2": (1.0, 2.0, 0.0)}
	for nm, (alpha, tau, ls) in cases.items():
		teacher_logits = 3.0 * torch.randn(6, 10, generator=g)
		class Teacher(nn.Module):
			def forward(self, x):
				return teacher_logits
		dummy = nn.Linear(1, 1)
		tr = DeiTTrainer(dummy, Teacher(), torch.optim.SGD(dummy.parameters(), lr=0.1), nn.CrossEntropyLoss(label_smoothing=ls), "cpu", alpha=alpha, tau=tau)
		s = (2.0 * torch.randn(6, 10, generator=g)).requires_grad_(True)
		y = torch.randint(0, 10, (6,), generator=g)
		loss = tr.compute_loss(s, y, torch.zeros(6, 3, 4, 4))
		loss.backward()
		arrays[f"{nm}/student"] = np_(s); arrays[f"{nm}/teacher"] = np_(teacher_logits); arrays[f"{nm}/y"] = np_(y)
		arrays[f"{nm}/loss"] = np_(loss); arrays[f"{nm}/dstudent"] = np_(s.grad)
	save(args.out, "deit_loss", arrays, {"kind": "deit_loss", "cases": {k: list(v) for k, v in cases.items()}})

	# AdamW (main.py:100 uses torch.optim.AdamW)
	arrays = {}
	p = torch.randn(257, generator=g).requires_grad_(True)
	opt = torch.optim.AdamW([p], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
	arrays["p0"] = np_(p)
	for step in range(3):
		gr = torch.randn(257, generator=g)
		p.grad = gr.clone()
		opt.step()
		arrays[f"g{step}"] = np_(gr); arrays[f"p{step + 1}"] = np_(p)
	save(args.out, "adamw", arrays, {"kind": "adamw", "lr": 1e-2, "betas": [0.9, 0.999], "eps": 1e-8, "weight_decay": 0.05})


if __name__ == "__main__":
	main()
