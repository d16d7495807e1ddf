"""CPU: the oracle restatement reproduces the golden vectors recorded from the reference modules.

Tolerance: 1e-4 relative (max|a-b| / max|b|) in fp32, the bar the north star sets for fp32."""
import pytest
import torch

from conftest import load_golden, relerr, sub
from oracle import vit_oracle as O

TOL = 1e-4


@pytest.mark.parametrize("name", ["vit_ln_gelu", "vit_p4_n65", "vit_rms_swiglu_ls", "vit_p16_geglu"])
def test_vit_model(name):
	g, meta = load_golden(name)
	c = meta["config"]
	sd = sub(g, "sd")
	fwd = lambda x, s: O.vit_fwd(x, s, c["n_patch"], c["n_block"], c["n_head"], c["norm"], c["activation"])
	logits, loss, grads, gx = O.loss_and_grads(fwd, sd, g["x"], g["y"])
	assert relerr(O.patchify(g["x"], c["n_patch"]), g["patches"]) == 0.0
	assert relerr(logits, g["logits"]) < TOL
	assert relerr(loss, g["loss"]) < TOL
	assert relerr(gx, g["grad_x"]) < TOL
	gg = sub(g, "grad")
	assert set(gg) == set(grads)
	for k in gg:
		assert relerr(grads[k], gg[k]) < TOL, k
	assert torch.equal(logits.argmax(-1), g["logits"].argmax(-1))


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_cvt_model(mode):
	g, meta = load_golden("cvt_small_" + mode)
	c = meta["config"]
	sd = sub(g, "sd")
	fwd = lambda x, s: O.cvt_fwd(x, s, c, training=(mode == "train"))
	logits, loss, grads, gx = O.loss_and_grads(fwd, sd, g["x"], g["y"])
	assert relerr(logits, g["logits"]) < TOL
	assert relerr(loss, g["loss"]) < TOL
	assert relerr(gx, g["grad_x"]) < 5 * TOL
	gg = sub(g, "grad")
	assert set(gg) == set(grads)
	for k in gg:
		assert relerr(grads[k], gg[k]) < 5 * TOL, k


@pytest.mark.parametrize("act", ["swiglu", "silu", "geglu", "gelu", "reglu", "relu", "reglu6", "relu6"])
def test_feedforward(act):
	g, _ = load_golden("feedforward")
	c = sub(g, act)
	sd = sub(c, "sd")
	x = c["x"].clone().requires_grad_(True)
	leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
	y = O.ffn_fwd(x, leaf, "", act)
	assert y.shape == c["y"].shape
	assert relerr(y, c["y"]) < TOL
	y.backward(c["dy"])
	assert relerr(x.grad, c["dx"]) < TOL
	for k, v in sub(c, "grad").items():
		assert relerr(leaf[k].grad, v) < TOL, k
	# explicit activation derivative against autograd of the explicit forward
	u = torch.linspace(-7, 7, 1001, dtype=torch.float64, requires_grad=True)
	kind = O.base_act(act)
	O.act_fwd(u, kind).sum().backward()
	assert relerr(O.act_grad(u.detach(), kind), u.grad) < 1e-12


def test_norms():
	g, _ = load_golden("norms")
	c = sub(g, "layernorm")
	y, mean, rstd = O.layernorm_fwd(c["x"], c["sd/weight"], c["sd/bias"])
	assert relerr(y, c["y"]) < TOL
	dx, dg, db = O.layernorm_bwd(c["dy"], c["x"], c["sd/weight"], mean, rstd)
	assert relerr(dx, c["dx"]) < TOL and relerr(dg, c["grad/weight"]) < TOL and relerr(db, c["grad/bias"]) < TOL
	for nm, off in (("rmsnorm", True), ("rmsnorm_nooffset", False)):
		c = sub(g, nm)
		y, rstd = O.rmsnorm_fwd(c["x"], c["sd/weight"], 1e-6, off)
		assert relerr(y, c["y"]) < TOL
		dx, dw = O.rmsnorm_bwd(c["dy"], c["x"], c["sd/weight"], rstd, off)
		assert relerr(dx, c["dx"]) < TOL and relerr(dw, c["grad/weight"]) < TOL


def test_dyt():
	"""DyT forward and explicit backward against the reference module's own outputs and autograd gradients."""
	g, _ = load_golden("dyt")
	c = sub(g, "mod")
	al, ga, be = c["sd/alpha"], c["sd/gamma"], c["sd/beta"]
	assert relerr(O.dyt_fwd(c["x"], al, ga, be), c["y"]) < TOL
	dx, dal, dga, dbe = O.dyt_bwd(c["dy"], c["x"], al, ga)
	assert relerr(dx, c["dx"]) < TOL and relerr(dal, c["grad/alpha"]) < TOL and relerr(dga, c["grad/gamma"]) < TOL and relerr(dbe, c["grad/beta"]) < TOL


@pytest.mark.parametrize("case", ["dh64", "dh12", "dh24", "dh64_n65"])
def test_mha(case):
	g, meta = load_golden("mha")
	c = sub(g, case)
	d, h, s = meta["cases"][case]
	sd = sub(c, "sd")
	x = c["x"].clone().requires_grad_(True)
	leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
	y = O.mha_fwd(x, leaf, "", h)
	assert relerr(y, c["y"]) < TOL
	y.backward(c["dy"])
	assert relerr(x.grad, c["dx"]) < TOL
	for k, v in sub(c, "grad").items():
		assert relerr(leaf[k].grad, v) < TOL, k


def test_sdpa_explicit_backward():
	torch.manual_seed(0)
	q = torch.randn(2, 3, 9, 16, dtype=torch.float64, requires_grad=True)
	k = torch.randn(2, 3, 5, 16, dtype=torch.float64, requires_grad=True)
	v = torch.randn(2, 3, 5, 16, dtype=torch.float64, requires_grad=True)
	out, lse = O.sdpa_fwd(q, k, v)
	ref = torch.nn.functional.scaled_dot_product_attention(q, k, v)
	assert relerr(out, ref) < 1e-12
	do = torch.randn_like(out)
	out.backward(do)
	dq, dk, dv = O.sdpa_bwd(do, q.detach(), k.detach(), v.detach(), out.detach(), lse.detach())
	assert relerr(dq, q.grad) < 1e-12 and relerr(dk, k.grad) < 1e-12 and relerr(dv, v.grad) < 1e-12


@pytest.mark.parametrize("case", ["s1", "s2", "s2_odd"])
def test_conv_attention(case):
	g, meta = load_golden("conv_attention")
	c = sub(g, case)
	dim, h, hw, stride = meta["cases"][case]
	sd = sub(c, "sd")
	x = c["x"].clone().requires_grad_(True)
	leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
	y = O.conv_attention_fwd(x, hw, hw, leaf, "", h, stride)
	assert relerr(y, c["y"]) < TOL
	y.backward(c["dy"])
	assert relerr(x.grad, c["dx"]) < TOL
	for k, v in sub(c, "grad").items():
		assert relerr(leaf[k].grad, v) < TOL, k


@pytest.mark.parametrize("case", ["k7s4p2", "k3s2p1"])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_conv_embedding(case, mode):
	g, meta = load_golden("conv_embedding")
	c = sub(g, f"{case}/{mode}")
	ci, co, k, s, p, hw = meta["cases"][case]
	sd = sub(c, "sd")
	x = c["x"].clone().requires_grad_(True)
	leaf = {kk: v.clone().requires_grad_(v.is_floating_point() and "running" not in kk) for kk, v in sd.items()}
	y = O.conv_embedding_fwd(x, leaf, "", s, p, training=(mode == "train"))
	assert relerr(y, c["y"]) < TOL
	y.backward(c["dy"])
	assert relerr(x.grad, c["dx"]) < 2 * TOL
	for kk, v in sub(c, "grad").items():
		assert relerr(leaf[kk].grad, v) < 2 * TOL, kk
	if mode == "train":
		conv = O.conv2d_fwd(c["x"], sd["conv.weight"], s, p)
		_, rm, rv = O.batchnorm2d_fwd(conv, sd["norm.weight"], sd["norm.bias"], sd["norm.running_mean"], sd["norm.running_var"], True)
		assert relerr(rm, c["sd_after/norm.running_mean"]) < TOL
		assert relerr(rv, c["sd_after/norm.running_var"]) < TOL


@pytest.mark.parametrize("case", ["a05_t1", "a03_t3_ls01", "a1_t2"])
def test_deit_loss(case):
	g, meta = load_golden("deit_loss")
	c = sub(g, case)
	alpha, tau, ls = meta["cases"][case]
	loss = O.deit_loss_fwd(c["student"], c["y"], c["teacher"], alpha, tau, ls)
	assert relerr(loss, c["loss"]) < TOL
	assert relerr(O.deit_loss_bwd(c["student"], c["y"], c["teacher"], alpha, tau, ls), c["dstudent"]) < TOL


def test_adamw():
	g, meta = load_golden("adamw")
	p = g["p0"]
	m = torch.zeros_like(p)
	v = torch.zeros_like(p)
	for step in range(3):
		p, m, v = O.adamw_step(p, g[f"g{step}"], m, v, step + 1, meta["lr"], *meta["betas"], meta["eps"], meta["weight_decay"])
		assert relerr(p, g[f"p{step + 1}"]) < 1e-5


@pytest.mark.parametrize("case", ["res_silu", "down_gelu", "wide_relu"])
@pytest.mark.parametrize("mode", ["train", "eval"])
def test_mbconv(case, mode):
	"""MBConv (layers/conv.py:5-44), incl. its activation quirk ("gelu" runs as ReLU) and the residual rule."""
	g, meta = load_golden("mbconv")
	c = sub(g, f"{case}/{mode}")
	ci, co, stride, expand, act, hw = meta["cases"][case]
	sd = sub(c, "sd")
	x = c["x"].clone().requires_grad_(True)
	leaf = {kk: v.clone().requires_grad_(v.is_floating_point() and "running" not in kk) for kk, v in sd.items()}
	y = O.mbconv_fwd(x, leaf, "", stride, act, training=(mode == "train"))
	assert relerr(y, c["y"]) < TOL
	y.backward(c["dy"])
	assert relerr(x.grad, c["dx"]) < 2 * TOL
	for kk, v in sub(c, "grad").items():
		assert relerr(leaf[kk].grad, v) < 2 * TOL, kk


def test_dual_head_loss_reduces_to_the_pinned_deit_loss():
	"""The two-head distillation loss with both heads fed the same logits is the reference's DeiTTrainer.compute_loss
	(pinned by deit_loss.npz); the distillation token itself has no reference counterpart (parity unpinned)."""
	g, meta = load_golden("deit_loss")
	for c, (alpha, tau, ls) in meta["cases"].items():
		d = sub(g, c)
		got = O.deit_dual_loss_fwd(d["student"], d["student"], d["y"], d["teacher"], alpha=alpha, tau=tau, label_smoothing=ls)
		assert relerr(got, d["loss"]) < TOL, c


def test_teacher_resnet20_matches_the_reference_teacher():
	"""The build-side CIFAR ResNet-20 (stock PyTorch modules) loads the reference teacher's state_dict strictly and
	reproduces its eval-mode logits (golden: the reference's models/resnet.py with its own checkpoint weights)."""
	from vit_amd.models.teacher import frozen_teacher
	g, meta = load_golden("resnet20_teacher")
	m = frozen_teacher(meta["depth"], meta["num_classes"], sub(g, "sd"))
	with torch.no_grad():
		assert relerr(m(g["x"]), g["logits"]) < TOL
	assert not any(p.requires_grad for p in m.parameters()) and not m.training
	# with its frozen BatchNorms folded into the convolutions (the form the distillation benchmark runs): same logits
	f = frozen_teacher(meta["depth"], meta["num_classes"], sub(g, "sd"), fold_bn=True)
	assert not any(isinstance(mod, torch.nn.BatchNorm2d) for mod in f.modules())
	with torch.no_grad():
		assert relerr(f(g["x"]), g["logits"]) < TOL
	assert not any(p.requires_grad for p in f.parameters()) and not f.training
