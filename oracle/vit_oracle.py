"""CPU oracle for the ViT / CvT forward+backward hot path.

TEST INFRASTRUCTURE ONLY.  This file restates, in explicit tensor arithmetic on
the CPU, what the reference's PyTorch modules compute on the hot path.  Only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it; the product (`vit_amd`) never does.

Every forward below is written out from the formulas (no `nn.Module`, no
`F.scaled_dot_product_attention`, no `F.layer_norm`, no einops); every leaf op
also has a hand-derived backward, which is what the HIP kernels implement.
Whole-model gradients are obtained by running torch autograd over the explicit
forward.  Parity of this restatement with the reference is pinned by the golden
vectors in `tests/golden/` (made by `tools/gen_golden.py`, which imports the
reference's own modules and records inputs, state_dict, logits, loss and every
gradient).

Citations are file:line under the reference checkout.
"""
import math

import torch

SQRT1_2 = 0.7071067811865476
INV_SQRT_2PI = 0.3989422804014327


# --------------------------------------------------------------------------
# leaf ops: forward + explicit backward
# --------------------------------------------------------------------------

def patchify(images, n_patch):
	"""layers/embeddings.py:7-16: 'b c (p1 hp) (p2 wp) -> b (p1 p2) (c hp wp)'."""
	b, c, h, w = images.shape
	assert h == w, "Patchify method is implemented for square images only"
	hp, wp = h // n_patch, w // n_patch
	x = images.reshape(b, c, n_patch, hp, n_patch, wp)
	x = x.permute(0, 2, 4, 1, 3, 5)  # b p1 p2 c hp wp
	return x.reshape(b, n_patch * n_patch, c * hp * wp)


def unpatchify(patches, n_patch, c):
	"""Inverse permutation of `patchify` (its backward)."""
	b, pp, k = patches.shape
	hp = wp = int(round(math.sqrt(k // c)))
	x = patches.reshape(b, n_patch, n_patch, c, hp, wp).permute(0, 3, 1, 4, 2, 5)
	return x.reshape(b, c, n_patch * hp, n_patch * wp)


def linear_fwd(x, w, b=None):
	"""nn.Linear: y = x @ w.T + b (models/vit.py:30,39; layers/attention.py:17-18;
	layers/transformers.py:36-40)."""
	y = x @ w.t()
	if b is not None:
		y = y + b
	return y


def linear_bwd(dy, x, w, has_bias=True):
	x2 = x.reshape(-1, x.shape[-1])
	dy2 = dy.reshape(-1, dy.shape[-1])
	dx = (dy2 @ w).reshape(x.shape)
	dw = dy2.t() @ x2
	db = dy2.sum(0) if has_bias else None
	return dx, dw, db


def layernorm_fwd(x, gamma, beta, eps=1e-5):
	"""nn.LayerNorm injected via config.norm (layers/vit.py:15,20,38,49)."""
	mean = x.mean(-1, keepdim=True)
	var = ((x - mean) ** 2).mean(-1, keepdim=True)
	rstd = torch.rsqrt(var + eps)
	xhat = (x - mean) * rstd
	return xhat * gamma + beta, mean.squeeze(-1), rstd.squeeze(-1)


def layernorm_bwd(dy, x, gamma, mean, rstd):
	d = x.shape[-1]
	xhat = (x - mean.unsqueeze(-1)) * rstd.unsqueeze(-1)
	g = dy * gamma
	c1 = g.sum(-1, keepdim=True) / d
	c2 = (g * xhat).sum(-1, keepdim=True) / d
	dx = (g - c1 - xhat * c2) * rstd.unsqueeze(-1)
	dgamma = (dy * xhat).reshape(-1, d).sum(0)
	dbeta = dy.reshape(-1, d).sum(0)
	return dx, dgamma, dbeta


def rmsnorm_fwd(x, weight, eps=1e-6, add_unit_offset=True):
	"""layers/norm.py:19-31: fp32 internal, y = x*rsqrt(mean(x^2)+eps)*(1+w), cast back."""
	xf = x.float()
	rstd = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
	scale = (1 + weight.float()) if add_unit_offset else weight.float()
	return (xf * rstd * scale).type_as(x), rstd.squeeze(-1)


def rmsnorm_bwd(dy, x, weight, rstd, add_unit_offset=True):
	d = x.shape[-1]
	xf, dyf = x.float(), dy.float()
	scale = (1 + weight.float()) if add_unit_offset else weight.float()
	r = rstd.unsqueeze(-1)
	xhat = xf * r
	g = dyf * scale
	c2 = (g * xhat).sum(-1, keepdim=True) / d
	dx = (g - xhat * c2) * r
	dw = (dyf * xhat).reshape(-1, d).sum(0)
	return dx.type_as(x), dw


def dyt_fwd(x, alpha, gamma, beta):
	"""layers/norm.py:42-45: gamma * tanh(alpha * x) + beta."""
	return gamma * torch.tanh(alpha * x) + beta


def dyt_bwd(dy, x, alpha, gamma):
	"""Gradients of dyt_fwd: (dx, dalpha [same shape as alpha], dgamma, dbeta)."""
	t = torch.tanh(alpha * x)
	u = dy * gamma * (1.0 - t * t)
	d = x.shape[-1]
	return u * alpha, (u * x).sum().reshape(alpha.shape), (dy * t).reshape(-1, d).sum(0), dy.reshape(-1, d).sum(0)


def act_fwd(u, kind):
	"""layers/transformers.py:26-33: F.gelu is the exact erf form."""
	if kind == "gelu":
		return 0.5 * u * (1.0 + torch.erf(u * SQRT1_2))
	if kind == "silu":
		return u * torch.sigmoid(u)
	if kind == "relu":
		return torch.clamp_min(u, 0.0)
	if kind == "relu6":
		return torch.clamp(u, 0.0, 6.0)
	if kind == "none":
		return u
	raise ValueError(kind)


def act_grad(u, kind):
	"""d act(u) / du."""
	if kind == "gelu":
		return 0.5 * (1.0 + torch.erf(u * SQRT1_2)) + u * INV_SQRT_2PI * torch.exp(-0.5 * u * u)
	if kind == "silu":
		s = torch.sigmoid(u)
		return s * (1.0 + u * (1.0 - s))
	if kind == "relu":
		return (u > 0).to(u.dtype)
	if kind == "relu6":
		return ((u > 0) & (u < 6)).to(u.dtype)
	if kind == "none":
		return torch.ones_like(u)
	raise ValueError(kind)


def base_act(activation):
	"""Which elementwise function an activation name selects
	(layers/transformers.py:26-33, substring tests in that order)."""
	a = activation.lower()
	if "silu" in a or "swiglu" in a:
		return "silu"
	if "gelu" in a or "geglu" in a:
		return "gelu"
	if "relu6" in a or "reglu6" in a:
		return "relu6"
	if "relu" in a or "reglu" in a:
		return "relu"
	raise ValueError(f"Unknown activation type: {activation}")


def is_gated(activation):
	"""Only swiglu/geglu double fc_in and gate (layers/transformers.py:35,48);
	reglu/reglu6 are listed as gated at :24 but behave as plain relu/relu6."""
	return activation.lower() in ("swiglu", "geglu")


def sdpa_fwd(q, k, v, scale=None):
	"""F.scaled_dot_product_attention(q,k,v,is_causal=False), dropout 0
	(layers/attention.py:39-43,84-88).  q:[B,h,Sq,dh] k,v:[B,h,Skv,dh].
	Returns (out, lse) with lse = logsumexp of the scaled scores per query row."""
	dh = q.shape[-1]
	scale = (1.0 / math.sqrt(dh)) if scale is None else scale
	s = (q @ k.transpose(-1, -2)) * scale
	m = s.max(-1, keepdim=True).values
	p = torch.exp(s - m)
	l = p.sum(-1, keepdim=True)
	out = (p / l) @ v
	lse = (m + torch.log(l)).squeeze(-1)
	return out, lse


def sdpa_bwd(do, q, k, v, out, lse, scale=None):
	dh = q.shape[-1]
	scale = (1.0 / math.sqrt(dh)) if scale is None else scale
	s = (q @ k.transpose(-1, -2)) * scale
	p = torch.exp(s - lse.unsqueeze(-1))
	dv = p.transpose(-1, -2) @ do
	dp = do @ v.transpose(-1, -2)
	delta = (do * out).sum(-1, keepdim=True)
	ds = p * (dp - delta) * scale
	dq = ds @ k
	dk = ds.transpose(-1, -2) @ q
	return dq, dk, dv


def _hash32(x):
	"""32-bit integer mixer of the attention dropout mask (numpy uint64 arithmetic masked to 32 bits)."""
	import numpy as np
	m = np.uint64(0xFFFFFFFF)
	x = x & m
	x ^= x >> np.uint64(16)
	x = (x * np.uint64(0x7feb352d)) & m
	x ^= x >> np.uint64(15)
	x = (x * np.uint64(0x846ca68b)) & m
	x ^= x >> np.uint64(16)
	return x


def _splitmix64(z):
	"""The 64-bit seed mixer applied on the host before the mask hash (plain Python integers modulo 2^64)."""
	m = (1 << 64) - 1
	z = (z + 0x9E3779B97F4A7C15) & m
	z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
	z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
	return z ^ (z >> 31)


def dropout_multiplier(seed, p, n, epoch=0):
	"""nn.Dropout (layers/vit.py:17-23, layers/transformers.py:20-21) as the HIP kernels draw it, for a contiguous tensor of
	n elements: float64 vector of 0 or 1 / (1 - p'), p' = p quantised to 1/65536.  Elements 2j and 2j+1 take the low and
	high 16 bits of one 32-bit hash of (seed, j).  The streaming dropout kernels and the GEMM epilogues use this same
	function (element index = row * N + column).  torch's Philox stream cannot be reproduced, so the tests hand THIS
	mask to the reference arithmetic.  epoch: value of the registered dropout-epoch word (a replayed HIP graph advances
	it per step so that every replay draws new masks from the seeds it was captured with); 0 leaves the seed alone."""
	import numpy as np
	if p <= 0.0:
		return torch.ones(n, dtype=torch.float64)
	m = np.uint64(0xFFFFFFFF)
	z = _splitmix64(int(seed) & ((1 << 64) - 1))
	k0, k1 = np.uint64(z & 0xFFFFFFFF), np.uint64(((z >> 32) + int(epoch) * 0x85EBCA6B) & 0xFFFFFFFF)
	t = float(np.float32(p) * np.float32(65536.0) + np.float32(0.5))
	thr = 65535 if t >= 65535.0 else (1 if t < 1.0 else int(t))
	pair = np.arange((n + 1) // 2, dtype=np.uint64)
	x = (pair & m) ^ k0
	x ^= x >> np.uint64(16)
	x = (x * np.uint64(0x7feb352d)) & m
	x = (x + k1 + (pair >> np.uint64(32)) * np.uint64(0x9E3779B1)) & m
	x ^= x >> np.uint64(15)
	x = (x * np.uint64(0x846ca68b)) & m
	x ^= x >> np.uint64(16)
	bits = np.stack([x & np.uint64(0xFFFF), x >> np.uint64(16)], axis=1).reshape(-1)[:n]
	keep = (bits >= np.uint64(thr)).astype(np.float64) * (65536.0 / (65536 - thr))
	return torch.from_numpy(keep)


def attn_dropout_keep(seed, p, B, H, Sq, Skv, epoch=0):
	"""Keep mask of the attention-probability dropout (the `dropout_p` of F.scaled_dot_product_attention,
	layers/attention.py:41,86), as the HIP kernels generate it: values 0 or 1/(1 - p'), shape [B, H, Sq, Skv], with
	p' = p quantised to 1/65536.  The reference draws its mask from torch's Philox stream, which cannot be reproduced;
	the tests therefore hand THIS mask to the reference arithmetic below (distribution parity, exact arithmetic parity).
	epoch: as in dropout_multiplier."""
	import numpy as np
	m = np.uint64(0xFFFFFFFF)
	thr = min(int(round(p * 65536.0)), 65535)
	if p > 0 and thr == 0:
		thr = 1
	seed_lo, seed_hi = np.uint64(((seed & 0xFFFFFFFF) + int(epoch) * 0x85EBCA6B) & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
	bh = np.arange(B * H, dtype=np.uint64)
	head_key = _hash32(seed_lo ^ _hash32(seed_hi ^ bh))  # [B*H]
	q = np.arange(Sq, dtype=np.uint64)
	row_key = _hash32((head_key[:, None] + q[None, :] * np.uint64(0x9E3779B1)) & m)  # [B*H, Sq]
	k = np.arange(Skv, dtype=np.uint64)
	pair = ((k >> np.uint64(1)) * np.uint64(0x85EBCA6B) + np.uint64(0x1B873593)) & m
	bits = _hash32(row_key[:, :, None] ^ pair[None, None, :])  # [B*H, Sq, Skv]
	field = np.where((k & np.uint64(1)) == 1, bits >> np.uint64(16), bits & np.uint64(0xFFFF))
	keep = (field >= np.uint64(thr)).astype(np.float64) * (65536.0 / (65536 - thr))
	return torch.from_numpy(keep.reshape(B, H, Sq, Skv))


def sdpa_dropout_fwd(q, k, v, keep, scale=None):
	"""sdpa_fwd with dropout on the probabilities: out = (softmax(s) * keep) @ v; lse is that of the full softmax."""
	dh = q.shape[-1]
	scale = (1.0 / math.sqrt(dh)) if scale is None else scale
	s = (q @ k.transpose(-1, -2)) * scale
	m = s.max(-1, keepdim=True).values
	p = torch.exp(s - m)
	l = p.sum(-1, keepdim=True)
	out = ((p / l) * keep.to(p.dtype)) @ v
	lse = (m + torch.log(l)).squeeze(-1)
	return out, lse


def sdpa_dropout_bwd(do, q, k, v, out, lse, keep, scale=None):
	dh = q.shape[-1]
	scale = (1.0 / math.sqrt(dh)) if scale is None else scale
	keep = keep.to(q.dtype)
	s = (q @ k.transpose(-1, -2)) * scale
	p = torch.exp(s - lse.unsqueeze(-1))
	dv = (p * keep).transpose(-1, -2) @ do
	dp = (do @ v.transpose(-1, -2)) * keep
	delta = (do * out).sum(-1, keepdim=True)
	ds = p * (dp - delta) * scale
	dq = ds @ k
	dk = ds.transpose(-1, -2) @ q
	return dq, dk, dv


def split_heads(x, h):
	"""'b s (h d) -> b h s d' (layers/attention.py:31-33,80-82)."""
	b, s, hd = x.shape
	return x.reshape(b, s, h, hd // h).permute(0, 2, 1, 3)


def merge_heads(x):
	"""'b h s d -> b s (h d)' (layers/attention.py:45,90)."""
	b, h, s, d = x.shape
	return x.permute(0, 2, 1, 3).reshape(b, s, h * d)


def cross_entropy_fwd(logits, y, label_smoothing=0.0):
	"""nn.CrossEntropyLoss(label_smoothing) mean reduction (main.py:106-109)."""
	lf = logits.float()
	logp = lf - torch.logsumexp(lf, -1, keepdim=True)
	nll = -logp.gather(1, y.view(-1, 1)).squeeze(1)
	if label_smoothing > 0:
		smooth = -logp.mean(-1)
		nll = (1 - label_smoothing) * nll + label_smoothing * smooth
	return nll.mean()


def cross_entropy_bwd(logits, y, label_smoothing=0.0):
	"""d mean-CE / d logits."""
	lf = logits.float()
	b, c = lf.shape
	p = torch.softmax(lf, -1)
	t = torch.zeros_like(p).scatter_(1, y.view(-1, 1), 1.0)
	if label_smoothing > 0:
		t = t * (1 - label_smoothing) + label_smoothing / c
	return (p - t) / b


def deit_loss_fwd(student, y, teacher, alpha=0.5, tau=1.0, label_smoothing=0.0):
	"""deit_trainer.py:30-45: (1-a)*CE(s,y) + a*tau^2*KL(log_softmax(s/tau) || softmax(t/tau)), batchmean."""
	base = cross_entropy_fwd(student, y, label_smoothing)
	ls = torch.log_softmax(student.float() / tau, 1)
	lt = torch.log_softmax(teacher.float() / tau, 1)
	pt = torch.exp(lt)
	kl = (pt * (lt - ls)).sum() / student.shape[0]
	return (1 - alpha) * base + alpha * kl * (tau * tau)


def deit_loss_bwd(student, y, teacher, alpha=0.5, tau=1.0, label_smoothing=0.0):
	"""d deit_loss / d student logits."""
	b = student.shape[0]
	g_ce = cross_entropy_bwd(student, y, label_smoothing)
	ps = torch.softmax(student.float() / tau, 1)
	pt = torch.softmax(teacher.float() / tau, 1)
	g_kl = (ps - pt) / (tau * b) * (tau * tau)
	return (1 - alpha) * g_ce + alpha * g_kl


def conv2d_fwd(x, w, stride, padding, groups=1):
	"""nn.Conv2d(bias=False) written as unfold + matmul
	(layers/embeddings.py:21, layers/attention.py:61-62)."""
	b, c, h, ww = x.shape
	co, cig, kh, kw = w.shape
	ho = (h + 2 * padding - kh) // stride + 1
	wo = (ww + 2 * padding - kw) // stride + 1
	cols = torch.nn.functional.unfold(x, (kh, kw), padding=padding, stride=stride)  # [b, c*kh*kw, ho*wo]
	if groups == 1:
		y = w.reshape(co, -1) @ cols
	else:
		assert groups == c and co == c and cig == 1
		cols = cols.reshape(b, c, kh * kw, ho * wo)
		y = (cols * w.reshape(1, c, kh * kw, 1)).sum(2)
	return y.reshape(b, co, ho, wo)


def batchnorm2d_fwd(x, gamma, beta, running_mean, running_var, training, eps=1e-5, momentum=0.1):
	"""nn.BatchNorm2d (layers/embeddings.py:22).  Returns (y, new_running_mean, new_running_var)."""
	if training:
		mean = x.mean((0, 2, 3))
		var = ((x - mean.view(1, -1, 1, 1)) ** 2).mean((0, 2, 3))
		n = x.numel() // x.shape[1]
		new_rm = (1 - momentum) * running_mean + momentum * mean
		new_rv = (1 - momentum) * running_var + momentum * var * n / max(n - 1, 1)
	else:
		mean, var = running_mean, running_var
		new_rm, new_rv = running_mean, running_var
	y = (x - mean.view(1, -1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1) + eps)
	return y * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1), new_rm, new_rv


# --------------------------------------------------------------------------
# composite forwards (autograd runs over these for whole-model gradients)
# --------------------------------------------------------------------------

def norm_apply(x, sd, prefix, kind):
	if kind == "layernorm":
		return layernorm_fwd(x, sd[prefix + "weight"], sd[prefix + "bias"], 1e-5)[0]
	if kind == "rmsnorm":
		return rmsnorm_fwd(x, sd[prefix + "weight"], 1e-6, True)[0]
	raise ValueError(kind)


def mha_fwd(x, sd, prefix, n_head):
	"""layers/attention.py:27-48 (MHA, use_qk_norm=False, eval / dropout 0)."""
	qkv = linear_fwd(x, sd[prefix + "qkv_proj.weight"])
	d = x.shape[-1]
	q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
	o, _ = sdpa_fwd(split_heads(q, n_head), split_heads(k, n_head), split_heads(v, n_head))
	return linear_fwd(merge_heads(o), sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def ffn_fwd(x, sd, prefix, activation):
	"""layers/transformers.py:45-57 (dropout 0)."""
	u = linear_fwd(x, sd[prefix + "fc_in.weight"], sd[prefix + "fc_in.bias"])
	kind = base_act(activation)
	if is_gated(activation):
		f = u.shape[-1] // 2
		gate, up = u[..., :f], u[..., f:]
		a = gate * act_fwd(up, kind)
	else:
		a = act_fwd(u, kind)
	return linear_fwd(a, sd[prefix + "fc_out.weight"], sd[prefix + "fc_out.bias"])


def residual(x, branch, sd, key):
	"""layers/vit.py:44-47,52-55."""
	if key in sd:
		return x + sd[key] * branch
	return x + branch


def vit_block_fwd(x, sd, prefix, n_head, norm, activation):
	"""layers/vit.py:27-57 for MHA attention."""
	a = mha_fwd(norm_apply(x, sd, prefix + "attn_norm.", norm), sd, prefix + "attn.", n_head)
	x = residual(x, a, sd, prefix + "attn_layer_scale")
	f = ffn_fwd(norm_apply(x, sd, prefix + "ff_norm.", norm), sd, prefix + "ff.", activation)
	return residual(x, f, sd, prefix + "ff_layer_scale")


def vit_fwd(images, sd, n_patch, n_block, n_head, norm="layernorm", activation="gelu"):
	"""models/vit.py:43-69 (MHA path): patchify -> linear_mapper -> prepend class_token ->
	+pos_embed -> blocks -> row 0 -> mlp.  No final norm."""
	b = images.shape[0]
	tok = linear_fwd(patchify(images, n_patch), sd["linear_mapper.weight"], sd["linear_mapper.bias"])
	cls = sd["class_token"].unsqueeze(0).expand(b, -1, -1)
	x = torch.cat([cls, tok], 1) + sd["pos_embed"]
	for i in range(n_block):
		x = vit_block_fwd(x, sd, f"blocks.{i}.", n_head, norm, activation)
	return linear_fwd(x[:, 0], sd["mlp.weight"], sd["mlp.bias"])


def deit_vit_fwd(images, sd, n_patch, n_block, n_head, norm="layernorm", activation="gelu"):
	"""The ViT of models/vit.py:43-69 with a second learned token (`dist_token`, row 1) and a second head (`dist_head`)
	that reads it: the student shape DeiT-style distillation uses.  PARITY UNPINNED for the token itself (the reference
	has no such model; an old trainer snapshot, _single_file.py:909-910, only accepts the output pair); every operation in
	here is one the golden-pinned vit_fwd uses.  Returns (class logits, distillation logits)."""
	b = images.shape[0]
	tok = linear_fwd(patchify(images, n_patch), sd["linear_mapper.weight"], sd["linear_mapper.bias"])
	cls = sd["class_token"].unsqueeze(0).expand(b, -1, -1)
	dist = sd["dist_token"].unsqueeze(0).expand(b, -1, -1)
	x = torch.cat([cls, dist, tok], 1) + sd["pos_embed"]
	for i in range(n_block):
		x = vit_block_fwd(x, sd, f"blocks.{i}.", n_head, norm, activation)
	return linear_fwd(x[:, 0], sd["mlp.weight"], sd["mlp.bias"]), linear_fwd(x[:, 1], sd["dist_head.weight"], sd["dist_head.bias"])


def deit_dual_loss_fwd(cls_logits, dist_logits, y, teacher, alpha=0.5, tau=1.0, label_smoothing=0.0):
	"""The two terms of deit_trainer.py:30-45 on two heads: hard-label cross entropy on the class head, the soft
	(teacher) term on the distillation head.  With cls_logits is dist_logits this is deit_loss_fwd, which the reference's
	DeiTTrainer.compute_loss golden pins."""
	base = cross_entropy_fwd(cls_logits, y, label_smoothing)
	ls = torch.log_softmax(dist_logits.float() / tau, 1)
	lt = torch.log_softmax(teacher.float() / tau, 1)
	kl = (torch.exp(lt) * (lt - ls)).sum() / dist_logits.shape[0]
	return (1 - alpha) * base + alpha * kl * (tau * tau)


def conv_embedding_fwd(x, sd, prefix, stride, padding, training=True):
	"""layers/embeddings.py:24-28: conv(bias=False) -> BatchNorm2d -> 'b c h w -> b (h w) c'."""
	y = conv2d_fwd(x, sd[prefix + "conv.weight"], stride, padding)
	y, _, _ = batchnorm2d_fwd(y, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"],
		sd[prefix + "norm.running_mean"], sd[prefix + "norm.running_var"], training)
	b, c, h, w = y.shape
	return y.reshape(b, c, h * w).transpose(1, 2)


def mbconv_act(activation):
	"""The activation MBConv really applies (layers/conv.py:12-17): the `if gelu / if silu / else` chain there has no
	`elif`, so "gelu" falls through to the final `else` and becomes ReLU; only "silu" is kept."""
	return "silu" if activation.lower() == "silu" else "relu"


def mbconv_fwd(x, sd, prefix, stride, activation, training=True):
	"""MBConv (layers/conv.py:20-44): Conv2d 1x1 -> BatchNorm2d -> act -> depthwise Conv2d 3x3 (stride, pad 1) ->
	BatchNorm2d -> act -> Conv2d 1x1 -> BatchNorm2d, plus the input when stride == 1 and the channel counts agree.
	x: [B, C, H, W].  State-dict keys are those of its nn.Sequential: conv.0 / conv.1 / conv.3 / conv.4 / conv.6 / conv.7."""
	act = mbconv_act(activation)
	c = prefix + "conv."

	def bn(y, i):
		return batchnorm2d_fwd(y, sd[f"{c}{i}.weight"], sd[f"{c}{i}.bias"], sd[f"{c}{i}.running_mean"], sd[f"{c}{i}.running_var"], training)[0]
	h = conv2d_fwd(x, sd[c + "0.weight"], 1, 0)
	h = act_fwd(bn(h, 1), act)
	h = conv2d_fwd(h, sd[c + "3.weight"], stride, 1, groups=h.shape[1])
	h = act_fwd(bn(h, 4), act)
	h = conv2d_fwd(h, sd[c + "6.weight"], 1, 0)
	h = bn(h, 7)
	if stride == 1 and x.shape[1] == h.shape[1]:
		h = x + h
	return h


def coatnet_fwd(x, sd, cfg, training=True):
	"""The conv + attention hybrid of _single_file.py:399-458 (the constructible older text of models/coatnet.py) composed
	from the pieces above, with the stem's convolution bias left out (it cancels in the BatchNorm behind it).  cfg: dims,
	depths, num_heads, activation, norm.  PARITY UNPINNED as a whole (the reference's class cannot be constructed); MBConv
	and the transformer block are pinned by their own goldens."""
	act = cfg["activation"]
	h = conv2d_fwd(x, sd["s0.conv.weight"], 2, 1)
	h, _, _ = batchnorm2d_fwd(h, sd["s0.norm.weight"], sd["s0.norm.bias"], sd["s0.norm.running_mean"], sd["s0.norm.running_var"], training)
	h = act_fwd(h, "gelu" if act.lower() == "gelu" else "silu")
	for stage, depth in (("s1", cfg["depths"][0]), ("s2", cfg["depths"][1])):
		for i in range(depth):
			h = mbconv_fwd(h, sd, f"{stage}.{i}.", 2 if i == 0 else 1, act, training)
	b, c, hh, ww = h.shape
	t = h.reshape(b, c, hh * ww).transpose(1, 2)
	t = linear_fwd(t, sd["s3_pe.weight"].reshape(sd["s3_pe.weight"].shape[0], -1), sd["s3_pe.bias"])
	for i in range(cfg["depths"][2]):
		t = vit_block_fwd(t, sd, f"s3.{i}.", cfg["num_heads"], cfg["norm"], act)
	h = conv2d_fwd(tokens_to_nchw(t, hh, ww), sd["s4_pe.weight"], 2, 0) + sd["s4_pe.bias"].view(1, -1, 1, 1)
	b, c, hh, ww = h.shape
	t = h.reshape(b, c, hh * ww).transpose(1, 2)
	for i in range(cfg["depths"][3]):
		t = vit_block_fwd(t, sd, f"s4.{i}.", cfg["num_heads"], cfg["norm"], act)
	return linear_fwd(t.mean(1), sd["head.weight"], sd["head.bias"])


def tokens_to_nchw(x, h, w):
	b, n, c = x.shape
	return x.transpose(1, 2).reshape(b, c, h, w)


def conv_attention_fwd(x, h, w, sd, prefix, n_head, stride, padding=1):
	"""layers/attention.py:67-93."""
	q = linear_fwd(x, sd[prefix + "q_proj.weight"], sd[prefix + "q_proj.bias"])
	k = linear_fwd(x, sd[prefix + "k_proj.weight"], sd[prefix + "k_proj.bias"])
	v = linear_fwd(x, sd[prefix + "v_proj.weight"], sd[prefix + "v_proj.bias"])
	c = x.shape[-1]
	k = conv2d_fwd(tokens_to_nchw(k, h, w), sd[prefix + "conv_proj_k.weight"], stride, padding, groups=c)
	v = conv2d_fwd(tokens_to_nchw(v, h, w), sd[prefix + "conv_proj_v.weight"], stride, padding, groups=c)
	b = x.shape[0]
	k = k.reshape(b, c, -1).transpose(1, 2)
	v = v.reshape(b, c, -1).transpose(1, 2)
	o, _ = sdpa_fwd(split_heads(q, n_head), split_heads(k, n_head), split_heads(v, n_head))
	return linear_fwd(merge_heads(o), sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def cvt_block_fwd(x, h, w, sd, prefix, n_head, stride, norm, activation):
	"""layers/cvt.py:32-50."""
	a = conv_attention_fwd(norm_apply(x, sd, prefix + "attn_norm.", norm), h, w, sd, prefix + "attn.", n_head, stride)
	x = residual(x, a, sd, prefix + "attn_layer_scale")
	f = ffn_fwd(norm_apply(x, sd, prefix + "ff_norm.", norm), sd, prefix + "ff.", activation)
	return residual(x, f, sd, prefix + "ff_layer_scale")


def cvt_fwd(x, sd, cfg, training=True):
	"""models/cvt.py:81-108.  cfg: dict with chw, s{1,2,3}_{emb_stride,emb_pad,depth,heads,qkv_stride}, norm, activation."""
	h, w = cfg["chw"][1], cfg["chw"][2]
	for s in (1, 2, 3):
		if s > 1:
			x = tokens_to_nchw(x, h, w)
		x = conv_embedding_fwd(x, sd, f"stage{s}_embedding.", cfg[f"s{s}_emb_stride"], cfg[f"s{s}_emb_pad"], training)
		h, w = h // cfg[f"s{s}_emb_stride"], w // cfg[f"s{s}_emb_stride"]
		for i in range(cfg[f"s{s}_depth"]):
			x = cvt_block_fwd(x, h, w, sd, f"stage{s}_blocks.{i}.", cfg[f"s{s}_heads"], cfg[f"s{s}_qkv_stride"], cfg["norm"], cfg["activation"])
	x = x.mean(1)
	x = norm_apply(x, sd, "norm.", cfg["norm"])
	return linear_fwd(x, sd["head.weight"], sd["head.bias"])


def loss_and_grads(fwd, sd, images, labels, loss_fn=cross_entropy_fwd):
	"""Run `fwd(images, sd)` with autograd over the explicit formulas above."""
	leaf = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
	x = images.detach().clone().requires_grad_(True)
	logits = fwd(x, leaf)
	loss = loss_fn(logits, labels)
	loss.backward()
	grads = {k: v.grad for k, v in leaf.items() if v.requires_grad and v.grad is not None}
	return logits.detach(), loss.detach(), grads, x.grad


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01):
	"""torch.optim.AdamW single-tensor update (main.py:100); step counts from 1."""
	p = p * (1 - lr * weight_decay)
	m = beta1 * m + (1 - beta1) * g
	v = beta2 * v + (1 - beta2) * g * g
	bc1 = 1 - beta1 ** step
	bc2 = 1 - beta2 ** step
	denom = v.sqrt() / math.sqrt(bc2) + eps
	p = p - (lr / bc1) * m / denom
	return p, m, v
