"""GPU parity sweep over seeded random shapes: the fixed cases of test_gpu_kernels.py pin the shapes of the models; this
file walks the dispatch boundaries between them (tile-size thresholds, vector-width fallbacks, ragged edges, the
persistent kernel's tile walk, K splits) with shapes nobody picked by hand.  Same oracle and tolerances."""
import os
import random

import pytest
import torch

from conftest import relerr
from oracle import vit_oracle as O

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}
SCALE = int(os.environ.get("VITK_SWEEP_SCALE", "1"))  # more cases per test for a longer hunt


def rnd(shape, dtype, dev, scale=1.0, seed=0):
	g = torch.Generator().manual_seed(seed)
	x = (torch.randn(shape, generator=g) * scale).to(dtype)
	return x.to(dev), x.double()


def _gemm_shapes(n, seed):
	r = random.Random(seed)
	out = []
	for _ in range(n):
		kind = r.choice(["small", "edge256", "many_tiles", "deep_k", "odd"])
		if kind == "small":
			M, N, K = r.randint(1, 300), r.randint(1, 300), r.randint(1, 200)
		elif kind == "edge256":  # around the 128/256 tile thresholds, aligned and not
			M = r.choice([255, 256, 257, 511, 512, 520]) + r.choice([0, 0, 1, 8])
			N = r.choice([248, 256, 264, 504, 512, 768])
			K = r.choice([56, 64, 72, 128, 192, 320])
		elif kind == "many_tiles":  # more 256x256 tiles than CUs: persistent walk with a partly filled last round
			M = r.randint(4200, 9000)
			N = r.choice([768, 1024, 1032, 2304])
			K = r.choice([64, 128, 256])
		elif kind == "deep_k":
			M, N, K = r.choice([256, 384, 768]), r.choice([256, 768]), r.choice([1536, 2048, 4160])
		else:
			M, N, K = r.randint(2, 700) | 1, r.randint(2, 500) | 1, r.randint(2, 300) | 1
		out.append((M, N, K))
	return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", range(12 * SCALE))
def test_gemm_random_shapes(gpu, dtype, case):
	from vit_amd import ops
	import vit_amd._lib as L
	M, N, K = _gemm_shapes(12 * SCALE, 2024)[case]
	if dtype == torch.float32 and M * N * K > 3e8:
		M = max(1, M // 16)  # the float32 kernel is the fallback path: keep its cases small
	a, ar = rnd((M, K), dtype, gpu, seed=case)
	b, br = rnd((N, K), dtype, gpu, 0.1, seed=100 + case)
	bias = torch.randn(N, generator=torch.Generator().manual_seed(case))
	pre_ref = ar @ br.t() + bias.double()
	y, pre = ops.linear_fwd(a, b, bias.to(gpu), act=L.ACT_CODES["silu"], save_pre=True)
	assert relerr(pre, pre_ref) < TOL[dtype], (M, N, K)
	assert relerr(y, O.act_fwd(pre.cpu().double(), "silu")) < TOL[dtype], (M, N, K)
	res, rr = rnd((M, N), dtype, gpu, seed=200 + case)
	y2 = ops.linear_fwd(a, b, bias.to(gpu), residual=res)
	assert relerr(y2, rr + pre_ref) < TOL[dtype], (M, N, K)
	# input gradient (weights [N, K] read MN-major, or through the transposed copy for large M) and weight gradient
	dy, dyr = rnd((M, N), dtype, gpu, seed=300 + case)
	dx = ops.linear_dgrad(dy, b)
	assert relerr(dx, dyr @ br) < TOL[dtype], (M, N, K)
	dw, db = ops.linear_wgrad(dy, a, bias_grad=True)
	wtol = 1e-4 if dtype == torch.float32 else 1e-3
	assert relerr(dw, dyr.t() @ ar) < wtol, (M, N, K)
	assert relerr(db, dyr.sum(0)) < wtol, (M, N, K)
	ops.weights_changed()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", range(10 * SCALE))
def test_attention_random_shapes(gpu, dtype, case):
	from vit_amd import ops
	r = random.Random(77 + case)
	dh = r.choice([64, 64, 64, 32, 24, 12])
	h = r.randint(1, 6)
	B = r.randint(1, 5)
	Sq = r.randint(1, 256)
	Skv = Sq if r.random() < 0.5 else r.randint(1, 256)
	d = h * dh
	q, qr = rnd((B * Sq, d), dtype, gpu, seed=case)
	k, kr = rnd((B * Skv, d), dtype, gpu, seed=50 + case)
	v, vr = rnd((B * Skv, d), dtype, gpu, seed=90 + case)
	do, dor = rnd((B * Sq, d), dtype, gpu, seed=130 + case)
	out, lse = ops.attention_fwd(q, k, v, B, h, Sq, Skv, dh)
	q4 = O.split_heads(qr.reshape(B, Sq, d), h)
	k4 = O.split_heads(kr.reshape(B, Skv, d), h)
	v4 = O.split_heads(vr.reshape(B, Skv, d), h)
	o_ref, lse_ref = O.sdpa_fwd(q4, k4, v4)
	tag = (B, h, Sq, Skv, dh)
	assert relerr(out, O.merge_heads(o_ref).reshape(B * Sq, d)) < TOL[dtype], tag
	assert relerr(lse, lse_ref) < 1e-4, tag
	dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
	ops.attention_bwd(do, q, k, v, out, lse, dq, dk, dv, B, h, Sq, Skv, dh)
	dq_r, dk_r, dv_r = O.sdpa_bwd(O.split_heads(dor.reshape(B, Sq, d), h), q4, k4, v4, o_ref, lse_ref)
	if Skv == 1:  # softmax over one key is the constant 1: dq, dk are zero up to rounding
		noise = 1e-4 if dtype == torch.float32 else 5e-2
		assert dq.abs().max().item() < noise and dk.abs().max().item() < noise, tag
	else:
		assert relerr(dq, O.merge_heads(dq_r).reshape(B * Sq, d)) < TOL[dtype], tag
		assert relerr(dk, O.merge_heads(dk_r).reshape(B * Skv, d)) < TOL[dtype], tag
	assert relerr(dv, O.merge_heads(dv_r).reshape(B * Skv, d)) < TOL[dtype], tag


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["layernorm", "rmsnorm"])
def test_norm_random_shapes(gpu, dtype, kind):
	from vit_amd import ops
	r = random.Random(5)
	for case in range(10 * SCALE):
		rows = r.randint(1, 3000)
		d = r.choice([r.randint(1, 100), 8 * r.randint(1, 512), 4 * r.randint(1, 200), 64 * r.randint(1, 64), 2 * r.randint(1, 512) + 1])
		x, xr = rnd((rows, d), dtype, gpu, seed=case)
		dy, dyr = rnd((rows, d), dtype, gpu, seed=20 + case)
		w = torch.randn(d, generator=torch.Generator().manual_seed(case)) * 0.5
		b = torch.randn(d, generator=torch.Generator().manual_seed(40 + case))
		eps = 1e-5 if kind == "layernorm" else 1e-6
		y, _, mean, rstd = ops.norm_fwd(x, w.to(gpu), b.to(gpu) if kind == "layernorm" else None, kind, eps, True)
		xs = x.cpu().double()
		if kind == "layernorm":
			y_ref, m_ref, r_ref = O.layernorm_fwd(xs, w.double(), b.double(), eps)
		else:
			y_ref, r_ref = O.rmsnorm_fwd(xs, w.double(), eps, True)
		assert relerr(y, y_ref) < TOL[dtype], (rows, d)
		dx, dw, db = ops.norm_bwd(dy, x, w.to(gpu), mean, rstd, kind, True)
		if kind == "layernorm":
			dx_ref, dw_ref, db_ref = O.layernorm_bwd(dyr, xs, w.double(), m_ref, r_ref)
			assert relerr(db, db_ref) < 1e-4, (rows, d)
		else:
			dx_ref, dw_ref = O.rmsnorm_bwd(dyr, xs, w.double(), r_ref, True)
		assert relerr(dx, dx_ref) < TOL[dtype], (rows, d)
		assert relerr(dw, dw_ref) < 1e-4, (rows, d)
