"""GPU parity: each vitk kernel (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (max|a-b| / max|b|): float32 1e-4; bfloat16 2e-2 (the bars the north star sets).
bf16 kernels are compared with the oracle evaluated in float64 on the SAME bf16-rounded inputs."""
import math

import pytest
import torch

from conftest import relerr
from oracle import vit_oracle as O

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}
DTYPES = [torch.float32, torch.bfloat16]


def rnd(shape, dtype, dev, scale=1.0, seed=0):
	g = torch.Generator().manual_seed(seed)
	x = (torch.randn(shape, generator=g) * scale).to(dtype)
	return x.to(dev), x.double()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (197, 384, 192), (300, 130, 72), (64, 1000, 768), (50, 10, 48), (512, 768, 3072), (33, 7, 5), (1000, 264, 192), (257, 1032, 64)])
@pytest.mark.parametrize("mode", ["fwd", "dgrad", "wgrad"])
def test_gemm_layouts(gpu, dtype, M, N, K, mode):
	from vit_amd import ops
	import vit_amd._lib as L
	if mode == "fwd":
		a, ar = rnd((M, K), dtype, gpu, seed=1)
		b, br = rnd((N, K), dtype, gpu, seed=2)
		c = torch.empty((M, N), dtype=dtype, device=gpu)
		ops.gemm(a, b, c, M, N, K, K, K, N)
		ref = ar @ br.t()
	elif mode == "dgrad":  # C[M,N] = A[M,K] @ W[K,N]  (B is MN-major)
		a, ar = rnd((M, K), dtype, gpu, seed=1)
		b, br = rnd((K, N), dtype, gpu, seed=2)
		c = torch.empty((M, N), dtype=dtype, device=gpu)
		ops.gemm(a, b, c, M, N, K, K, N, N, b_layout=L.MN_MAJOR)
		ref = ar @ br
	else:  # C[M,N] = A[K,M]^T @ B[K,N], float32 output
		a, ar = rnd((K, M), dtype, gpu, seed=1)
		b, br = rnd((K, N), dtype, gpu, seed=2)
		c = torch.empty((M, N), dtype=torch.float32, device=gpu)
		ops.gemm(a, b, c, M, N, K, M, N, N, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR)
		ref = ar.t() @ br
	tol = TOL[dtype] if (mode != "wgrad" or dtype == torch.float32) else 1e-3  # wgrad output is f32: only accumulation-order error
	assert relerr(c, ref) < tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_exact_integers(gpu, dtype):
	"""Small-integer operands: every product and sum is exact, so any fragment-layout or swizzle error shows as a wrong integer.
	A is the identity-like selector with an asymmetric B (catches transposed outputs)."""
	from vit_amd import ops
	import vit_amd._lib as L
	M, N, K = 256, 384, 320
	g = torch.Generator().manual_seed(3)
	a = torch.randint(-3, 4, (M, K), generator=g).to(dtype)
	b = torch.randint(-3, 4, (N, K), generator=g).to(dtype)
	c = torch.empty((M, N), dtype=dtype, device=gpu)
	ops.gemm(a.to(gpu), b.to(gpu), c, M, N, K, K, K, N)
	ref = a.double() @ b.double().t()
	assert torch.equal(c.cpu().double(), ref.to(dtype).double())  # exact sum, one rounding to the output type
	# MN-major operands (transposing LDS reads)
	at, bt = a.t().contiguous(), b.t().contiguous()
	c2 = torch.empty((M, N), dtype=torch.float32, device=gpu)
	ops.gemm(at.to(gpu), bt.to(gpu), c2, M, N, K, M, N, N, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR)
	assert torch.equal(c2.cpu().double(), ref)
	c3 = torch.empty((M, N), dtype=dtype, device=gpu)
	ops.gemm(a.to(gpu), bt.to(gpu), c3, M, N, K, K, N, N, b_layout=L.MN_MAJOR)
	assert torch.equal(c3.cpu().double(), ref.to(dtype).double())


@pytest.mark.parametrize("M,N,K", [(4096, 384, 1536), (4096, 1152, 384), (1000, 264, 192), (520, 136, 448), (128, 128, 192), (2048, 64, 256), (72, 1024, 1024)])
@pytest.mark.parametrize("a_mn,b_mn", [(False, False), (False, True), (True, True), (True, False)])
def test_gemm_few_tiles_long_k(gpu, M, N, K, a_mn, b_mn):
	"""The four-stage form of the 128x128 kernel (at most one workgroup per CU, K of three or more tiles: the GEMMs of a stage
	with few tokens), every pair of operand layouts, ragged last tiles: small-integer operands, so the float32 result is
	the exact sum and a wrong fragment, swizzle, stage or wait shows as a wrong integer; then the fused epilogues on it."""
	import vit_amd._lib as L
	from vit_amd import ops
	g = torch.Generator().manual_seed(M + N + K)
	a = torch.randint(-3, 4, (M, K), generator=g).to(torch.bfloat16)
	b = torch.randint(-3, 4, (N, K), generator=g).to(torch.bfloat16)
	ref = a.double() @ b.double().t()
	A = (a.t().contiguous() if a_mn else a).to(gpu)
	B = (b.t().contiguous() if b_mn else b).to(gpu)
	for out_dtype in (torch.float32, torch.bfloat16):
		c = torch.full((M, N), float("nan"), dtype=out_dtype, device=gpu)
		ops.gemm(A, B, c, M, N, K, M if a_mn else K, N if b_mn else K, N, a_layout=L.MN_MAJOR if a_mn else L.K_MAJOR, b_layout=L.MN_MAJOR if b_mn else L.K_MAJOR,
			workspace=False)
		assert torch.equal(c.cpu().double(), ref.to(out_dtype).double())
	if a_mn:
		return
	# bias + GELU with the pre-activation as second output; bias + LayerScale + residual; activation derivative
	x, xr = rnd((M, K), torch.bfloat16, gpu, seed=1)
	w, wr = rnd((N, K), torch.bfloat16, gpu, 0.1, seed=2)
	W = w.t().contiguous() if b_mn else w
	bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
	lay = dict(b_layout=L.MN_MAJOR if b_mn else L.K_MAJOR)
	ldb = N if b_mn else K
	pre_ref = xr @ wr.t() + bias.double()
	y = torch.empty((M, N), dtype=torch.bfloat16, device=gpu)
	pre = torch.empty_like(y)
	ops.gemm(x, W, y, M, N, K, K, ldb, N, bias=bias.to(gpu), epilogue=L.EPI_ACT, act=L.ACT_GELU, out2=pre, ld_out2=N, **lay)
	assert relerr(pre, pre_ref) < TOL[torch.bfloat16]
	assert relerr(y, O.act_fwd(pre.cpu().double(), "gelu")) < TOL[torch.bfloat16]
	res, rr = rnd((M, N), torch.bfloat16, gpu, seed=4)
	gamma = torch.rand(N, generator=torch.Generator().manual_seed(5)) + 0.5
	ops.gemm(x, W, y, M, N, K, K, ldb, N, bias=bias.to(gpu), epilogue=L.EPI_RESIDUAL, aux=res, ld_aux=N, gamma=gamma.to(gpu), **lay)
	assert relerr(y, rr + gamma.double() * pre_ref) < TOL[torch.bfloat16]
	ops.gemm(x, W, y, M, N, K, K, ldb, N, epilogue=L.EPI_ACTGRAD, act=L.ACT_GELU, aux=pre, ld_aux=N, **lay)
	assert relerr(y, (xr @ wr.t()) * O.act_grad(pre.cpu().double(), "gelu")) < TOL[torch.bfloat16]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", ["gelu", "silu", "relu", "relu6"])
def test_gemm_epilogues(gpu, dtype, act):
	from vit_amd import ops
	import vit_amd._lib as L
	M, N, K = 200, 256, 128
	x, xr = rnd((M, K), dtype, gpu, seed=1)
	w, wr = rnd((N, K), dtype, gpu, 0.1, seed=2)
	bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
	code = L.ACT_CODES[act]
	y, pre = ops.linear_fwd(x, w, bias.to(gpu), act=code, save_pre=True)
	pre_ref = xr @ wr.t() + bias.double()
	assert relerr(pre, pre_ref) < TOL[dtype]
	assert relerr(y, O.act_fwd(pre.cpu().double(), act)) < TOL[dtype]
	# residual + LayerScale epilogue
	res, rr = rnd((M, N), dtype, gpu, seed=4)
	gamma = torch.rand(N, generator=torch.Generator().manual_seed(5)) + 0.5
	y2 = ops.linear_fwd(x, w, bias.to(gpu), residual=res, gamma=gamma.to(gpu))
	assert relerr(y2, rr + gamma.double() * pre_ref) < TOL[dtype]
	# dgrad with the activation derivative in the epilogue
	dy, dyr = rnd((M, K), dtype, gpu, seed=6)          # gradient wrt the NEXT layer's input == this layer's activation output
	w2, w2r = rnd((K, N), dtype, gpu, 0.1, seed=7)     # next layer weight [out=K, in=N]
	du = ops.linear_dgrad(dy, w2, act=code, pre=pre)
	assert relerr(du, (dyr @ w2r) * O.act_grad(pre.cpu().double(), act)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,p", [(10, 0.3), (4097, 0.1), (200 * 256, 0.5), (7, 0.999)])
def test_dropout_mask_matches_oracle(gpu, dtype, n, p):
	"""The streaming dropout kernel draws exactly the mask the oracle restates (bit for bit), alone and stacked."""
	from vit_amd import ops
	x = torch.ones(n, device=gpu).to(dtype)
	got = ops.dropout_add(x, None, p, 1234567)
	want = O.dropout_multiplier(1234567, p, n)
	assert torch.equal(got.cpu().double() != 0, want != 0)
	assert relerr(got, want) < 5e-3 if dtype == torch.bfloat16 else relerr(got, want) < 1e-6
	kept = float((want != 0).double().mean())
	if n > 1000:
		assert abs(kept - (1 - p)) < 0.03
	got2 = ops.dropout_add(x, None, p, 1234567, 0.25, 99)
	assert torch.equal(got2.cpu().double() != 0, (want != 0) & (O.dropout_multiplier(99, 0.25, n) != 0))
	assert not torch.equal(O.dropout_multiplier(1234568, p, n) != 0, want != 0) or n < 16


@pytest.mark.parametrize("dtype", DTYPES)
def test_dropout_epoch(gpu, dtype):
	"""A registered dropout-epoch word is mixed into the seed of every mask at execution time (what lets a replayed HIP
	graph draw new masks from the seeds it was captured with): value 0 leaves the masks of the seeds alone; another value
	gives the masks the oracle restates for that epoch, the same in the streaming kernel, the GEMM epilogue (persistent
	bf16 kernel and generic kernel) and the attention kernels (whole-key-set, key-blocked and generic), forward and backward."""
	from vit_amd import ops
	n, p, seed = 4096, 0.3, 4242
	x = torch.ones(n, device=gpu).to(dtype)
	base = O.dropout_multiplier(seed, p, n)
	ep = ops.DropoutEpoch(gpu)
	try:
		assert torch.equal(ops.dropout_add(x, None, p, seed).cpu().double() != 0, base != 0)
		ep.advance()
		ep.advance()
		torch.cuda.synchronize()
		assert int(ep.word[0]) == 2
		want = O.dropout_multiplier(seed, p, n, epoch=2)
		assert not torch.equal(want != 0, base != 0)
		assert torch.equal(ops.dropout_add(x, None, p, seed).cpu().double() != 0, want != 0)
		for M, N, K in ((1024, 1024, 128), (33, 7, 5)):
			a, ar = rnd((M, K), dtype, gpu, seed=1)
			w, wr = rnd((N, K), dtype, gpu, 0.1, seed=2)
			res, rr = rnd((M, N), dtype, gpu, seed=4)
			m2 = O.dropout_multiplier(seed, p, M * N, epoch=2).reshape(M, N)
			y = ops.linear_fwd(a, w, None, residual=res, drop=((p, seed),))
			assert relerr(y, rr + (ar @ wr.t()) * m2) < TOL[dtype]
		for B, h, Sq, Skv, dh in ((2, 3, 65, 65, 64), (1, 2, 300, 77, 12)):
			d = h * dh
			q, qr = rnd((B * Sq, d), dtype, gpu, seed=1)
			k, kr = rnd((B * Skv, d), dtype, gpu, seed=2)
			v, vr = rnd((B * Skv, d), dtype, gpu, seed=3)
			do, dor = rnd((B * Sq, d), dtype, gpu, seed=4)
			keep = O.attn_dropout_keep(seed, p, B, h, Sq, Skv, epoch=2)
			assert not torch.equal(keep != 0, O.attn_dropout_keep(seed, p, B, h, Sq, Skv) != 0)
			out, lse = ops.attention_fwd(q, k, v, B, h, Sq, Skv, dh, drop=(p, seed))
			q4, k4, v4 = (O.split_heads(t.reshape(B, -1, d), h) for t in (qr, kr, vr))
			o_ref, lse_ref = O.sdpa_dropout_fwd(q4, k4, v4, keep)
			assert relerr(out, O.merge_heads(o_ref).reshape(B * Sq, d)) < TOL[dtype]
			dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
			ops.attention_bwd(do, q, k, v, out, lse, dq, dk, dv, B, h, Sq, Skv, dh, drop=(p, seed))
			dq_r, dk_r, dv_r = O.sdpa_dropout_bwd(O.split_heads(dor.reshape(B, Sq, d), h), q4, k4, v4, o_ref, lse_ref, keep)
			assert relerr(dq, O.merge_heads(dq_r).reshape(B * Sq, d)) < TOL[dtype]
			assert relerr(dk, O.merge_heads(dk_r).reshape(B * Skv, d)) < TOL[dtype]
			assert relerr(dv, O.merge_heads(dv_r).reshape(B * Skv, d)) < TOL[dtype]
	finally:
		ep.close()
	assert torch.equal(ops.dropout_add(x, None, p, seed).cpu().double() != 0, base != 0)  # unregistered: the seed alone again


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(200, 256, 128), (4363, 3848, 128), (1024, 1024, 128), (300, 130, 72), (33, 7, 5)])
def test_gemm_fused_dropout(gpu, dtype, M, N, K):
	"""Dropout fused into the GEMM epilogues (activation output, branch before the residual add, gradient in front of
	the activation derivative) uses the mask of the streaming kernel on element m * N + n: every kernel path (float32
	MFMA, 128x128, 256x256 single-tile and persistent with ragged edges, unaligned and odd N)."""
	from vit_amd import ops
	import vit_amd._lib as L
	if dtype == torch.float32 and M > 2000:
		pytest.skip("large shapes are the bf16 kernels' cases")
	x, xr = rnd((M, K), dtype, gpu, seed=1)
	w, wr = rnd((N, K), dtype, gpu, 0.1, seed=2)
	bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
	d1, d2 = (0.3, 4242), (0.1, 77)
	m1 = O.dropout_multiplier(d1[1], d1[0], M * N).reshape(M, N)
	m12 = m1 * O.dropout_multiplier(d2[1], d2[0], M * N).reshape(M, N)
	y, pre = ops.linear_fwd(x, w, bias.to(gpu), act=L.ACT_CODES["gelu"], save_pre=True, drop=(d1,))
	pre_ref = xr @ wr.t() + bias.double()
	assert relerr(pre, pre_ref) < TOL[dtype]
	assert torch.equal(y.cpu() == 0, (m1 == 0) | (y.cpu() == 0)) and relerr(y, O.act_fwd(pre.cpu().double(), "gelu") * m1) < TOL[dtype]
	res, rr = rnd((M, N), dtype, gpu, seed=4)
	gamma = torch.rand(N, generator=torch.Generator().manual_seed(5)) + 0.5
	y2 = ops.linear_fwd(x, w, bias.to(gpu), residual=res, gamma=gamma.to(gpu), drop=((0.0, 0), d1, d2)[1:])
	assert relerr(y2, rr + gamma.double() * (pre_ref * m12)) < TOL[dtype]
	y3 = ops.linear_fwd(x, w, bias.to(gpu), residual=res, drop=((0.0, 0), d2))  # an inactive first entry is skipped
	assert relerr(y3, rr + pre_ref * O.dropout_multiplier(d2[1], d2[0], M * N).reshape(M, N)) < TOL[dtype]
	dy, dyr = rnd((M, K), dtype, gpu, seed=6)
	w2, w2r = rnd((K, N), dtype, gpu, 0.1, seed=7)
	du = ops.linear_dgrad(dy, w2, act=L.ACT_CODES["gelu"], pre=pre, drop=(d1,))
	assert relerr(du, (dyr @ w2r) * m1 * O.act_grad(pre.cpu().double(), "gelu")) < TOL[dtype]
	with pytest.raises(Exception):
		ops.gemm(x, w, torch.empty((M, N), dtype=dtype, device=gpu), M, N, K, K, K, N, drop=(d1,))  # no epilogue to carry it


def test_gemm_row_split_last_round(gpu):
	"""N = 768, K >= 1536, 300 tiles of 256 x 256 on 256 CUs: the whole round runs on the persistent kernel, the 44
	remaining tiles' rows on the 128 x 128 kernel.  Both blocks against the float64 product, with every epilogue that
	carries a row-indexed operand (residual, pre-activation in, pre-activation out)."""
	from vit_amd import ops
	import vit_amd._lib as L
	dtype = torch.bfloat16
	M, N, K = 25600, 768, 1536
	x, xr = rnd((M, K), dtype, gpu, 0.5, seed=1)
	w, wr = rnd((N, K), dtype, gpu, 0.05, seed=2)
	bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
	res, rr = rnd((M, N), dtype, gpu, seed=4)
	ref = xr @ wr.t() + bias.double()
	for rows in (slice(0, M), slice(M - 3000, M)):  # the whole result, and the second block on its own
		y = ops.linear_fwd(x, w, bias.to(gpu), residual=res)
		assert relerr(y[rows], (rr + ref)[rows]) < TOL[dtype]
		a, pre = ops.linear_fwd(x, w, bias.to(gpu), act=L.ACT_CODES["gelu"], save_pre=True)
		assert relerr(pre[rows], ref[rows]) < TOL[dtype]
		assert relerr(a[rows], O.act_fwd(pre.cpu().double(), "gelu")[rows]) < TOL[dtype]
	dy, dyr = rnd((M, K), dtype, gpu, seed=6)   # gradient of a [M, K]-wide next layer whose weight is [K, N]
	w2, w2r = rnd((K, N), dtype, gpu, 0.05, seed=7)
	ops.weights_changed()
	du = ops.linear_dgrad(dy, w2, act=L.ACT_CODES["gelu"], pre=pre)
	want = (dyr @ w2r) * O.act_grad(pre.cpu().double(), "gelu")
	assert relerr(du, want) < TOL[dtype] and relerr(du[M - 3000:], want[M - 3000:]) < TOL[dtype]


def test_gemm_fused_dropout_across_the_row_split(gpu):
	"""N = 768, K >= 1536 with a partly filled last round of 256 x 256 tiles: the call runs as two row blocks on two
	kernels; the second block must draw the mask of ITS rows of the whole tensor (drop_row0), or the backward pass, which
	masks the gradient of the whole tensor, would disagree with the forward pass on those rows."""
	from vit_amd import ops
	dtype = torch.bfloat16
	M, N, K = 300 * 256 // 3, 768, 1536  # 100 x 3 tiles on 256 CUs: one whole round + 44 tiles
	x, _ = rnd((M, K), dtype, gpu, 0.5, seed=1)
	w, _ = rnd((N, K), dtype, gpu, 0.05, seed=2)
	res, rr = rnd((M, N), dtype, gpu, seed=4)
	d1 = (0.3, 991)
	plain = ops.linear_fwd(x, w, None).cpu().double()  # the same two kernels without dropout
	y = ops.linear_fwd(x, w, None, residual=res, drop=(d1,))
	m1 = O.dropout_multiplier(d1[1], d1[0], M * N).reshape(M, N)
	assert relerr(y, rr + plain * m1) < TOL[dtype]
	tail = slice(M - 2000, M)  # rows of the second block on their own: a mask restarted at row 0 would fail here
	assert relerr(y[tail], (rr + plain * m1)[tail]) < TOL[dtype]


@pytest.mark.parametrize("act", ["gelu", "silu"])
def test_gemm_more_tiles_than_cus(gpu, act):
	"""bf16 GEMMs with more 256x256 output tiles than the GPU has CUs (ragged last tile row and column): the
	persistent kernel walks several tiles per workgroup and writes the epilogue straight from the accumulators."""
	from vit_amd import ops
	import vit_amd._lib as L
	dtype = torch.bfloat16
	M, N, K = 4363, 3848, 128
	x, xr = rnd((M, K), dtype, gpu, seed=1)
	w, wr = rnd((N, K), dtype, gpu, 0.1, seed=2)
	bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
	code = L.ACT_CODES[act]
	plain = ops.linear_fwd(x, w, None)
	assert relerr(plain, xr @ wr.t()) < TOL[dtype]
	y, pre = ops.linear_fwd(x, w, bias.to(gpu), act=code, save_pre=True)
	pre_ref = xr @ wr.t() + bias.double()
	assert relerr(pre, pre_ref) < TOL[dtype]
	assert relerr(y, O.act_fwd(pre.cpu().double(), act)) < TOL[dtype]
	res, rr = rnd((M, N), dtype, gpu, seed=4)
	gamma = torch.rand(N, generator=torch.Generator().manual_seed(5)) + 0.5
	y2 = ops.linear_fwd(x, w, bias.to(gpu), residual=res, gamma=gamma.to(gpu))
	assert relerr(y2, rr + gamma.double() * pre_ref) < TOL[dtype]
	dy, dyr = rnd((M, K), dtype, gpu, seed=6)
	w2, w2r = rnd((K, N), dtype, gpu, 0.1, seed=7)
	dx = ops.linear_dgrad(dy, w2)
	assert relerr(dx, dyr @ w2r) < TOL[dtype]
	du = ops.linear_dgrad(dy, w2, act=code, pre=pre)
	assert relerr(du, (dyr @ w2r) * O.act_grad(pre.cpu().double(), act)) < TOL[dtype]
	# exact small integers through the same path: any tile-walk or fragment-placement error is a wrong integer
	g = torch.Generator().manual_seed(8)
	a = torch.randint(-3, 4, (M, K), generator=g).to(dtype)
	b = torch.randint(-3, 4, (N, K), generator=g).to(dtype)
	c = ops.linear_fwd(a.to(gpu), b.to(gpu), None)
	assert torch.equal(c.cpu().double(), (a.double() @ b.double().t()).to(dtype).double())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(6400, 768, 512), (197 * 64, 768, 768), (197 * 32, 2304, 768), (6400 + 64, 256, 1024), (4099 * 8, 264, 320), (1000, 264, 192), (300, 130, 72), (197 * 8, 1000, 768),
	(33280, 192, 576), (8000, 64, 192), (20000, 200, 136),  # these three: narrow layers over many rows (many K slices through scratch)
	(33280, 192, 192), (33280, 768, 192), (33792, 192, 768), (65536, 64, 256), (16384, 384, 192), (4096, 128, 64), (8192, 320, 448),  # the weight gradients of the 32 x 32 models (layer widths 64 ... 768 over 4 k ... 65 k tokens)
	(512, 10, 192), (256, 10, 384), (4100, 10, 64), (700, 3, 1000)])  # 10-class heads: one or a few output tiles of the float32-MFMA kernel, K (the batch) cut over workgroups
def test_wgrad_with_bias_grad(gpu, dtype, M, N, K):
	"""dw = dy^T x together with db = column sums of dy (what autograd's addmm backward produces), on every GEMM path:
	large tiles with split-K (column sums fused into the kernel; whole-tile shapes on the deep-pipeline kernel, also with a
	one-step last K slice), 128x128 tiles and the float32 kernel (column-sum
	kernel behind the GEMM)."""
	from vit_amd import ops
	dy, dyr = rnd((M, N), dtype, gpu, seed=1)
	x, xr = rnd((M, K), dtype, gpu, seed=2)
	dw, db = ops.linear_wgrad(dy, x, bias_grad=True)
	tol = 1e-4 if dtype == torch.float32 else 1e-3  # float32 outputs: accumulation-order error only
	assert relerr(dw, dyr.t() @ xr) < tol
	assert relerr(db, dyr.sum(0)) < tol
	assert relerr(ops.linear_wgrad(dy, x), dyr.t() @ xr) < tol  # and without the bias gradient
	# a caller without scratch memory: K slices are summed with float atomics into C; also accumulation into C
	import vit_amd._lib as L
	c = torch.full((N, K), 1.0, dtype=torch.float32, device=gpu)
	ops.gemm(dy, x, c, N, K, M, N, K, K, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR, workspace=False)
	assert relerr(c, dyr.t() @ xr) < tol
	ops.gemm(dy, x, c, N, K, M, N, K, K, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR, accumulate=True, workspace=False)
	ops.gemm(dy, x, c, N, K, M, N, K, K, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR, accumulate=True)
	assert relerr(c, 3.0 * (dyr.t() @ xr)) < tol


GROUPS = {
	# (tokens, dy width, x width, bias gradient, accumulate): the four weight gradients of a block
	"vit_tiny_block": [(33280, 576, 192, False, True), (33280, 192, 192, True, True), (33280, 768, 192, True, True), (33280, 192, 768, True, True)],
	"cvt_stage1_block": [(65536, 192, 64, True, True), (65536, 64, 64, True, False), (65536, 256, 64, True, True), (65536, 64, 256, True, False)],
	"cvt_stage3_block": [(4096, 1152, 384, True, True), (4096, 384, 384, True, True), (4096, 1536, 384, True, True), (4096, 384, 1536, True, True)],
	# token counts that differ (one K slice for the short one), ragged widths, more products than one launch takes
	"mixed_lengths": [(20000, 200, 136, True, True), (1024, 72, 264, True, False), (33280, 192, 576, False, False), (2048, 64, 64, True, True)],
	"twelve": [(8192 + 64 * i, 64 + 8 * i, 192, i % 2 == 0, i % 3 != 0) for i in range(12)],
	# more output tiles than the launch has workgroup slots (8 x 90 tiles of 128 x 128): one K slice each, several rounds
	"many_tiles": [(1024 + 64 * i, 1152, 1280, i % 2 == 0, i % 2 == 1) for i in range(8)],
	# products the grouped kernel does not take ride along (large tiles; few rows; float32 operands are a separate case below)
	"with_others": [(33280, 192, 192, True, True), (197 * 64, 2304, 768, True, True), (16384, 192, 576, True, False), (512, 10, 192, True, True)],
}


@pytest.mark.parametrize("name", sorted(GROUPS))
def test_gemm_group_matches_separate_launches(gpu, name):
	"""vitk_gemm_group: several weight-gradient products (dW = dy^T x, db = column sums of dy) in one call give what one
	vitk_gemm per product gives (float32 results, so only the summation order differs) and what float64 gives."""
	import vit_amd._lib as L
	from vit_amd import ops
	dtype = torch.bfloat16
	probs = []
	for i, (M, N, K, bias, acc) in enumerate(GROUPS[name]):
		dy, dyr = rnd((M, N), dtype, gpu, seed=10 + i)
		x, xr = rnd((M, K), dtype, gpu, seed=50 + i)
		probs.append((dy, x, dyr, xr, bias, acc))
	start = 0.5  # what the accumulating products add into
	def run(hold):
		outs = []
		for dy, x, _, _, bias, acc in probs:
			(M, N), K = dy.shape, x.shape[1]
			dw = torch.full((N, K), start, dtype=torch.float32, device=gpu) if acc else torch.full((N, K), float("nan"), dtype=torch.float32, device=gpu)
			db = torch.full((N,), start, dtype=torch.float32, device=gpu) if bias else None
			ops.gemm(dy, x, dw, N, K, M, N, K, K, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR, a_colsum=db, accumulate=acc, hold=hold)
			outs.append((dw, db))
		return outs
	single = run(None)
	held = []
	grouped = run(held)
	assert len(held) == len(probs)
	descs = (L.GemmDesc * len(held))(*[h[0] for h in held])
	L.check(L.load().vitk_gemm_group(descs, len(held), L.stream()))
	if name in ("vit_tiny_block", "cvt_stage1_block", "cvt_stage3_block", "mixed_lengths", "twelve", "many_tiles"):
		assert L.load().vitk_gemm_last_kernel().decode() == "gemm_bf16_group_kernel"
	for (dw1, db1), (dwg, dbg), (dy, x, dyr, xr, bias, acc) in zip(single, grouped, probs):
		ref = dyr.t() @ xr + (start if acc else 0.0)
		assert relerr(dwg, ref) < 1e-3
		assert relerr(dwg, dw1.double().cpu()) < 1e-5
		if bias:
			assert relerr(dbg, dyr.sum(0) + start) < 1e-3
			assert relerr(dbg, db1.double().cpu()) < 1e-5
	# nothing to do, one product, float32 operands (not the grouped kernel's): all through the same entry point
	L.check(L.load().vitk_gemm_group(None, 0, L.stream()))
	a, ar = rnd((300, 72), torch.float32, gpu, seed=3)
	b, br = rnd((300, 130), torch.float32, gpu, seed=4)
	held = []
	c = torch.empty((72, 130), dtype=torch.float32, device=gpu)
	ops.gemm(a, b, c, 72, 130, 300, 72, 130, 130, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR, hold=held)
	L.check(L.load().vitk_gemm_group((L.GemmDesc * 1)(held[0][0]), 1, L.stream()))
	assert relerr(c, ar.t() @ br) < 1e-4
	# a bad descriptor anywhere in the list: an error, and nothing is launched
	held = []
	c2 = torch.zeros((72, 130), dtype=torch.float32, device=gpu)
	ops.gemm(a, b, c2, 72, 130, 300, 72, 130, 130, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR, hold=held)
	ops.gemm(a, b, c2, 72, 130, 300, 72, 130, 130, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR, hold=held)
	held[1][0].ldc = 1
	assert L.load().vitk_gemm_group((L.GemmDesc * 2)(held[0][0], held[1][0]), 2, L.stream()) != 0
	torch.cuda.synchronize()
	assert float(c2.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,C,HW,P", [(3, 3, 32, 8), (2, 3, 224, 14), (5, 1, 48, 6), (2, 2, 12, 2), (1, 3, 30, 10)])
def test_patchify_kernels(gpu, dtype, B, C, HW, P):
	"""The patch gather and its inverse scatter (layers/embeddings.py:7-16), element-wise and 16-byte forms (patch edges
	4, 16, 8, 6, 3): bit-exact against the oracle, and the scatter undoes the gather."""
	from vit_amd import ops
	import vit_amd._lib as L
	x, _ = rnd((B, C, HW, HW), dtype, gpu, seed=21)
	K = C * (HW // P) ** 2
	patches = torch.empty((B * P * P, K), dtype=dtype, device=gpu)
	L.check(L.load().vitk_patchify(L.ptr(x), L.ptr(patches), B, C, HW, P, L.dtype_code(dtype), L.stream()))
	assert torch.equal(patches.cpu().reshape(B, P * P, K), O.patchify(x.cpu(), P))
	back = torch.zeros_like(x)
	L.check(L.load().vitk_unpatchify(L.ptr(patches), L.ptr(back), B, C, HW, P, L.dtype_code(dtype), L.stream()))
	assert torch.equal(back.cpu(), x.cpu())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,cols", [(768, 3072), (65, 130), (1, 7), (200, 64)])
def test_transpose(gpu, dtype, rows, cols):
	from vit_amd import ops
	x, _ = rnd((rows, cols), dtype, gpu, seed=11)
	y = ops.transpose(x)
	assert y.shape == (cols, rows) and torch.equal(y.cpu(), x.cpu().t().contiguous())  # a copy: bit-exact
	# strided source (a column block of a wider matrix)
	wide, _ = rnd((rows, cols + 8), dtype, gpu, seed=12)
	assert torch.equal(ops.transpose(wide[:, 3:3 + cols]).cpu(), wide.cpu()[:, 3:3 + cols].t().contiguous())


def test_dgrad_transposed_weight_path(gpu):
	"""Many-token bf16 input gradients run on a cached transposed weight copy: same result as the direct path, and the
	cache follows weight updates (version bump / weights_changed())."""
	from vit_amd import ops
	import vit_amd._lib as L
	M, N, K = 4500, 320, 384  # dy [M, N], w [N, K]
	dy, dyr = rnd((M, N), torch.bfloat16, gpu, seed=1)
	w, wr = rnd((N, K), torch.bfloat16, gpu, 0.1, seed=2)
	pre, _ = rnd((M, K), torch.bfloat16, gpu, seed=3)
	a = ops.linear_dgrad(dy, w)
	assert relerr(a, dyr @ wr) < TOL[torch.bfloat16]
	ops.DGRAD_TRANSPOSED_WEIGHTS = False
	try:
		b = ops.linear_dgrad(dy, w)
		b2 = ops.linear_dgrad(dy, w, act=L.ACT_GELU, pre=pre)
	finally:
		ops.DGRAD_TRANSPOSED_WEIGHTS = True
	assert relerr(a, b.cpu().double()) < 1e-2  # two kernels, same bf16 operands: accumulation order only
	a2 = ops.linear_dgrad(dy, w, act=L.ACT_GELU, pre=pre)
	assert relerr(a2, b2.cpu().double()) < 1e-2
	w.mul_(2.0)  # in-place update through torch bumps the version: the cached transpose must not be reused
	assert relerr(ops.linear_dgrad(dy, w), 2.0 * (dyr @ wr)) < TOL[torch.bfloat16]
	L.check(L.load().vitk_cast(L.ptr((wr * 3.0).to(torch.bfloat16).to(gpu)), L.ptr(w), w.numel(), L.BF16, L.BF16, L.stream()))  # behind torch's back
	ops.weights_changed()
	assert relerr(ops.linear_dgrad(dy, w), 3.0 * (dyr @ wr)) < TOL[torch.bfloat16]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kind", ["layernorm", "rmsnorm"])
@pytest.mark.parametrize("rows,d", [(37, 768), (9, 192), (5, 40), (3, 50), (11, 2048), (1, 768), (1030, 768), (9, 256), (6, 1280),
	(20001, 768), (40003, 192), (36000, 50)])  # the last three: more rows than waves in the grid (every wave loops, odd tails)
def test_norms(gpu, dtype, kind, rows, d):
	from vit_amd import ops
	x, xr = rnd((rows, d), dtype, gpu, 1.5, seed=1)
	res, rr = rnd((rows, d), dtype, gpu, seed=2)
	dy, dyr = rnd((rows, d), dtype, gpu, seed=3)
	dres, dresr = rnd((rows, d), dtype, gpu, seed=4)
	g = torch.Generator().manual_seed(5)
	w = torch.randn(d, generator=g) * 0.3 + (1.0 if kind == "layernorm" else 0.0)
	b = torch.randn(d, generator=g) * 0.3
	eps = 1e-5 if kind == "layernorm" else 1e-6
	y, s, mean, rstd = ops.norm_fwd(x, w.to(gpu), b.to(gpu) if kind == "layernorm" else None, kind, eps, True, res=res)
	s_ref = (xr + rr).to(dtype).double()
	assert relerr(s, s_ref) < 1e-6 if dtype == torch.float32 else relerr(s, s_ref) == 0.0
	if kind == "layernorm":
		y_ref, m_ref, r_ref = O.layernorm_fwd(s_ref, w.double(), b.double(), eps)
		assert relerr(mean, m_ref) < 1e-5
	else:
		y_ref, r_ref = O.rmsnorm_fwd(s_ref, w.double(), eps, True)
	assert relerr(rstd, r_ref) < 1e-5
	assert relerr(y, y_ref) < TOL[dtype]
	dx, dw, db = ops.norm_bwd(dy, s, w.to(gpu), mean, rstd, kind, True, dres=dres)
	if kind == "layernorm":
		dx_ref, dw_ref, db_ref = O.layernorm_bwd(dyr, s_ref, w.double(), m_ref, r_ref)
		assert relerr(db, db_ref) < 1e-4
	else:
		dx_ref, dw_ref = O.rmsnorm_bwd(dyr, s_ref, w.double(), r_ref, True)
	assert relerr(dx, dx_ref + dresr) < TOL[dtype]
	assert relerr(dw, dw_ref) < 1e-4
	# without the scratch buffer the column sums are added with float atomics: same results
	dx2, dw2, db2 = ops.norm_bwd(dy, s, w.to(gpu), mean, rstd, kind, True, dres=dres, workspace=False)
	assert torch.equal(dx2, dx) and relerr(dw2, dw_ref) < 1e-4
	if kind == "layernorm":
		assert relerr(db2, db_ref) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,h,Sq,Skv,dh", [(2, 3, 65, 65, 64), (2, 12, 197, 197, 64), (3, 4, 17, 17, 12), (2, 2, 50, 13, 24), (2, 6, 4, 1, 64), (1, 1, 300, 300, 64), (2, 2, 16, 4, 32),
	(70, 12, 65, 65, 64), (67, 9, 40, 24, 64),  # these two: more (batch, head) pairs than CUs (several pairs per workgroup, ragged last workgroup)
	# key-blocked online-softmax kernels (bf16): the conv + attention hybrid's stages, a CvT-sized first stage, ragged blocks, every head-size class
	(2, 4, 784, 784, 12), (2, 4, 196, 196, 24), (1, 2, 3136, 3136, 64), (1, 2, 300, 77, 40), (1, 3, 129, 64, 8), (1, 1, 64, 64, 4), (2, 2, 257, 513, 48), (1, 2, 5, 1, 20),
	(1, 2, 1, 130, 16), (1, 2, 260, 260, 64),
	(2, 3, 200, 250, 28), (1, 5, 600, 256, 12), (3, 2, 70, 129, 32)])  # one-pass backward of the key-blocked family (small heads, at most 256 keys): ragged key and query blocks
def test_attention(gpu, dtype, B, h, Sq, Skv, dh):
	from vit_amd import ops
	d = h * dh
	q, qr = rnd((B * Sq, d), dtype, gpu, seed=1)
	k, kr = rnd((B * Skv, d), dtype, gpu, seed=2)
	v, vr = rnd((B * Skv, d), dtype, gpu, seed=3)
	do, dor = rnd((B * Sq, d), dtype, gpu, seed=4)
	out, lse = ops.attention_fwd(q, k, v, B, h, Sq, Skv, dh)
	q4 = O.split_heads(qr.reshape(B, Sq, d), h)
	k4 = O.split_heads(kr.reshape(B, Skv, d), h)
	v4 = O.split_heads(vr.reshape(B, Skv, d), h)
	o_ref, lse_ref = O.sdpa_fwd(q4, k4, v4)
	assert relerr(out, O.merge_heads(o_ref).reshape(B * Sq, d)) < TOL[dtype]
	assert relerr(lse, lse_ref) < 1e-4
	dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
	ops.attention_bwd(do, q, k, v, out, lse, dq, dk, dv, B, h, Sq, Skv, dh)
	do4 = O.split_heads(dor.reshape(B, Sq, d), h)
	dq_r, dk_r, dv_r = O.sdpa_bwd(do4, q4, k4, v4, o_ref, lse_ref)
	if Skv == 1:
		# one key: softmax is the constant 1, dq and dk are exact zeros in exact arithmetic and rounding noise here
		noise = 1e-4 if dtype == torch.float32 else 5e-2
		assert dq.abs().max().item() < noise and dk.abs().max().item() < noise
	else:
		assert relerr(dq, O.merge_heads(dq_r).reshape(B * Sq, d)) < TOL[dtype]
		assert relerr(dk, O.merge_heads(dk_r).reshape(B * Skv, d)) < TOL[dtype]
	assert relerr(dv, O.merge_heads(dv_r).reshape(B * Skv, d)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,h,Sq,Skv,dh,p", [(2, 12, 197, 197, 64, 0.3), (2, 3, 65, 65, 64, 0.1), (2, 2, 50, 13, 24, 0.3), (1, 2, 300, 300, 64, 0.5), (2, 2, 64, 16, 32, 0.3),
	(70, 12, 65, 65, 64, 0.3), (67, 9, 40, 24, 64, 0.2),  # these two: more (batch, head) pairs than CUs, so a workgroup walks several pairs
	(2, 4, 784, 784, 12, 0.3), (2, 4, 196, 196, 24, 0.2), (1, 2, 300, 77, 40, 0.3), (1, 2, 520, 520, 64, 0.1), (1, 3, 129, 64, 8, 0.5), (2, 3, 200, 250, 28, 0.3)])  # key-blocked kernels
def test_attention_dropout(gpu, dtype, B, h, Sq, Skv, dh, p):
	"""Attention with dropout on the probabilities (the reference's training path, layers/attention.py:41,86): the
	kernels' mask is regenerated on the CPU and handed to the oracle, so forward and all three gradients are compared
	exactly as without dropout; the mask itself is checked for its keep rate."""
	from vit_amd import ops
	d = h * dh
	seed = 0x1234_5678_9ABC_DEF0 + Sq
	q, qr = rnd((B * Sq, d), dtype, gpu, seed=1)
	k, kr = rnd((B * Skv, d), dtype, gpu, seed=2)
	v, vr = rnd((B * Skv, d), dtype, gpu, seed=3)
	do, dor = rnd((B * Sq, d), dtype, gpu, seed=4)
	keep = O.attn_dropout_keep(seed, p, B, h, Sq, Skv)
	rate = (keep > 0).double().mean().item()
	assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / keep.numel()) ** 0.5 + 1e-4
	out, lse = ops.attention_fwd(q, k, v, B, h, Sq, Skv, dh, drop=(p, seed))
	q4 = O.split_heads(qr.reshape(B, Sq, d), h)
	k4 = O.split_heads(kr.reshape(B, Skv, d), h)
	v4 = O.split_heads(vr.reshape(B, Skv, d), h)
	o_ref, lse_ref = O.sdpa_dropout_fwd(q4, k4, v4, keep)
	assert relerr(out, O.merge_heads(o_ref).reshape(B * Sq, d)) < TOL[dtype]
	assert relerr(lse, lse_ref) < 1e-4
	# a different seed gives a different mask
	out2, _ = ops.attention_fwd(q, k, v, B, h, Sq, Skv, dh, drop=(p, seed + 1))
	assert relerr(out2, O.merge_heads(o_ref).reshape(B * Sq, d)) > 10 * TOL[dtype]
	dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
	ops.attention_bwd(do, q, k, v, out, lse, dq, dk, dv, B, h, Sq, Skv, dh, drop=(p, seed))
	do4 = O.split_heads(dor.reshape(B, Sq, d), h)
	dq_r, dk_r, dv_r = O.sdpa_dropout_bwd(do4, q4, k4, v4, o_ref, lse_ref, keep)
	assert relerr(dq, O.merge_heads(dq_r).reshape(B * Sq, d)) < TOL[dtype]
	assert relerr(dk, O.merge_heads(dk_r).reshape(B * Skv, d)) < TOL[dtype]
	assert relerr(dv, O.merge_heads(dv_r).reshape(B * Skv, d)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S,dh", [(33, 64), (90, 12), (300, 64)])
def test_attention_strided_qkv(gpu, dtype, S, dh):
	"""q/k/v as column slices of one fused [rows, 3d] buffer (the layout MHA uses)."""
	from vit_amd import ops
	B, h = 2, 2
	d = h * dh
	qkv, qkvr = rnd((B * S, 3 * d), dtype, gpu, seed=1)
	out, lse = ops.attention_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, h, S, S, dh)
	q4, k4, v4 = (O.split_heads(qkvr[:, i * d:(i + 1) * d].reshape(B, S, d), h) for i in range(3))
	o_ref, lse_ref = O.sdpa_fwd(q4, k4, v4)
	assert relerr(out, O.merge_heads(o_ref).reshape(B * S, d)) < TOL[dtype]
	do, dor = rnd((B * S, d), dtype, gpu, seed=4)
	dqkv = torch.empty_like(qkv)
	ops.attention_bwd(do, qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], out, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, h, S, S, dh)
	refs = O.sdpa_bwd(O.split_heads(dor.reshape(B, S, d), h), q4, k4, v4, o_ref, lse_ref)
	for i, r in enumerate(refs):
		assert relerr(dqkv[:, i * d:(i + 1) * d], O.merge_heads(r).reshape(B * S, d)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise(gpu, dtype):
	from vit_amd import ops
	import vit_amd._lib as L
	lib = L.load()
	x, xr = rnd((37, 96), dtype, gpu, 2.0, seed=1)
	# colsum
	assert relerr(ops.colsum(x), xr.sum(0)) < 1e-4
	assert relerr(ops.colsum(x[:, :10]), xr[:, :10].sum(0)) < 1e-4
	# gated act fwd/bwd
	for act in ("silu", "gelu"):
		code = L.ACT_CODES[act]
		a = ops.gated_act_fwd(x, 48, code)
		assert relerr(a, xr[:, :48] * O.act_fwd(xr[:, 48:], act)) < TOL[dtype]
		da, dar = rnd((37, 48), dtype, gpu, seed=2)
		du = ops.gated_act_bwd(da, x, 48, code)
		ref = torch.cat([dar * O.act_fwd(xr[:, 48:], act), dar * xr[:, :48] * O.act_grad(xr[:, 48:], act)], 1)
		assert relerr(du, ref) < TOL[dtype]
	# scale_residual + layerscale bwd
	res, rr = rnd((37, 96), dtype, gpu, seed=3)
	gamma = torch.rand(96, generator=torch.Generator().manual_seed(4)) + 0.5
	assert relerr(ops.scale_residual(x, res, gamma.to(gpu)), rr + gamma.double() * xr) < TOL[dtype]
	db, dg = ops.layerscale_bwd(res, x, gamma.to(gpu))
	assert relerr(db, gamma.double() * rr) < TOL[dtype]
	assert relerr(dg, (rr * xr).sum(0)) < 1e-4
	# cast round trip
	y = ops.cast(ops.cast(x, torch.float32), dtype)
	assert torch.equal(y, x)
	# dropout: keep fraction and scaling; backward regenerates the same mask
	big = torch.ones(1 << 20, dtype=dtype, device=gpu)
	out = ops.DropoutFn.apply(big, 0.3, 12345)
	keep = (out != 0).float().mean().item()
	assert abs(keep - 0.7) < 5e-3
	assert relerr(out[out != 0], torch.full((1,), 1 / 0.7).expand(int((out != 0).sum()))) < TOL[dtype]
	out2 = ops.DropoutFn.apply(big, 0.3, 12345)
	assert torch.equal(out, out2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["a05_t1", "a03_t3_ls01", "a1_t2"])
def test_ce_kd_loss_golden(gpu, dtype, case):
	from conftest import load_golden, sub
	from vit_amd import ops
	g, meta = load_golden("deit_loss")
	c = sub(g, case)
	alpha, tau, ls = meta["cases"][case]
	s = c["student"].to(dtype).to(gpu).requires_grad_(True)
	loss = ops.deit_loss(s, c["y"].to(gpu), c["teacher"].to(dtype).to(gpu), alpha, tau, ls)
	loss.backward()
	if dtype == torch.float32:
		assert relerr(loss, c["loss"]) < 1e-4
		assert relerr(s.grad, c["dstudent"]) < 1e-4
	else:
		ref_l = O.deit_loss_fwd(s.detach().cpu().double(), c["y"], c["teacher"].to(dtype).double(), alpha, tau, ls)
		assert relerr(loss, ref_l) < 1e-3
		assert relerr(s.grad, c["dstudent"]) < 2e-2
	# plain cross entropy
	s2 = c["student"].to(gpu).requires_grad_(True)
	l2 = ops.cross_entropy(s2, c["y"].to(gpu), ls)
	l2.backward()
	assert relerr(l2, O.cross_entropy_fwd(c["student"], c["y"], ls)) < 1e-4
	assert relerr(s2.grad, O.cross_entropy_bwd(c["student"], c["y"], ls)) < 1e-4


def test_adamw_golden(gpu):
	from conftest import load_golden
	import vit_amd._lib as L
	g, meta = load_golden("adamw")
	lib = L.load()
	p = g["p0"].to(gpu).clone()
	m = torch.zeros_like(p)
	v = torch.zeros_like(p)
	p16 = torch.empty(p.shape, dtype=torch.bfloat16, device=gpu)
	for step in range(3):
		gr = g[f"g{step}"].to(gpu)
		L.check(lib.vitk_adamw(L.ptr(p), L.ptr(gr), L.ptr(m), L.ptr(v), None, L.ptr(p16), p.numel(), meta["lr"], meta["betas"][0], meta["betas"][1],
			meta["eps"], meta["weight_decay"], step + 1, 1.0, L.stream()))
		assert relerr(p, g[f"p{step + 1}"]) < 1e-5
	assert torch.equal(p16, p.to(torch.bfloat16))


def test_streaming_kernels_at_sizes_that_loop(gpu):
	"""AdamW and dropout over several million elements (odd counts): the grids are capped, so every thread walks a
	stride loop that the small golden cases never enter.  AdamW against the update formula of torch.optim.AdamW
	(main.py:100 there) evaluated by the oracle in float64; the dropout mask against the oracle's, bit for bit."""
	import vit_amd._lib as L
	from vit_amd import ops
	lib = L.load()
	n = 6_000_003
	gen = torch.Generator().manual_seed(0)
	p0 = torch.randn(n, generator=gen)
	g0 = torch.randn(n, generator=gen) * 0.1
	m0 = torch.randn(n, generator=gen) * 0.01
	v0 = torch.rand(n, generator=gen) * 0.01
	mask = (torch.rand(n, generator=gen) < 0.7).to(torch.uint8)  # weight decay on 70 % of the elements
	p, g, m, v = (t.to(gpu).clone() for t in (p0, g0, m0, v0))
	p16 = torch.empty(n, dtype=torch.bfloat16, device=gpu)
	lr, b1, b2, eps, wd, step, gscale = 1e-3, 0.9, 0.999, 1e-8, 0.05, 7, 0.5
	L.check(lib.vitk_adamw(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(mask.to(gpu)), L.ptr(p16), n, lr, b1, b2, eps, wd, step, gscale, L.stream()))
	want_p = torch.empty(n, dtype=torch.float64)
	dec = mask.bool()
	for sel, w in ((dec, wd), (~dec, 0.0)):
		pp, mm, vv = O.adamw_step(p0[sel].double(), g0[sel].double() * gscale, m0[sel].double(), v0[sel].double(), step, lr, b1, b2, eps, w)
		want_p[sel] = pp
	assert relerr(p, want_p) < 1e-6
	assert torch.equal(p16, p.to(torch.bfloat16))
	# dropout over the same count: stride loop + the last, unpaired element
	x = torch.ones(n, device=gpu)
	got = ops.dropout_add(x, None, 0.3, 424242)
	assert torch.equal(got.cpu().double() != 0, O.dropout_multiplier(424242, 0.3, n) != 0)


@pytest.mark.parametrize("B,C", [(5000, 1000), (70000, 10), (33, 1001), (1, 7)])
def test_ce_kd_loss_shapes(gpu, B, C):
	"""The DeiT loss kernel (deit_trainer.py:23-52 there) beyond the golden cases: ImageNet's class count, more rows than
	the grid holds, a class count that is no multiple of the vector width, one row.  Float32 against the oracle."""
	from vit_amd import ops
	g = torch.Generator().manual_seed(B + C)
	s0 = torch.randn(B, C, generator=g) * 2.0
	t0 = torch.randn(B, C, generator=g) * 3.0
	y = torch.randint(0, C, (B,), generator=g)
	for alpha, tau, ls in ((0.5, 2.0, 0.1), (0.0, 1.0, 0.0)):
		s = s0.to(gpu).requires_grad_(True)
		loss = ops.deit_loss(s, y.to(gpu), t0.to(gpu), alpha, tau, ls)
		loss.backward()
		assert relerr(loss, O.deit_loss_fwd(s0.double(), y, t0.double(), alpha, tau, ls)) < 1e-4
		assert relerr(s.grad, O.deit_loss_bwd(s0.double(), y, t0.double(), alpha, tau, ls)) < 1e-4
	s = s0.to(gpu).requires_grad_(True)
	l2 = ops.cross_entropy(s, y.to(gpu), 0.1)
	l2.backward()
	assert relerr(l2, O.cross_entropy_fwd(s0.double(), y, 0.1)) < 1e-4
	assert relerr(s.grad, O.cross_entropy_bwd(s0.double(), y, 0.1)) < 1e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,tokens,d,C", [(512, 66, 192, 10), (37, 5, 192, 10), (64, 199, 768, 1000), (3, 2, 1100, 7), (1, 2, 64, 1)])
def test_dual_head_over_batch_slabs_and_classes(gpu, dtype, B, tokens, d, C):
	"""Classifier on token row 0 and distillation head on token row 1 by one launch (models/vit.py:63-65 there, twice): few
	classes spread the samples over many waves (10 classes at batch 512: slabs of 2 samples), 1000 classes keep a wave per
	class over the whole batch; ragged last slab, d beyond the register-resident 1024 columns."""
	import vit_amd._lib as L
	x, xr = rnd((B, tokens, d), dtype, gpu, seed=1)
	w0, w0r = rnd((C, d), dtype, gpu, scale=0.1, seed=2)
	w1, w1r = rnd((C, d), dtype, gpu, scale=0.1, seed=3)
	b0, b0r = rnd((C,), torch.float32, gpu, seed=4)
	b1, b1r = rnd((C,), torch.float32, gpu, seed=5)
	y0 = torch.full((B, C), 7.0, dtype=dtype, device=gpu)
	y1 = torch.full((B, C), 7.0, dtype=dtype, device=gpu)
	L.check(L.load().vitk_dual_head_fwd(L.ptr(x), L.ptr(w0), L.ptr(b0), L.ptr(w1), L.ptr(b1), L.ptr(y0), L.ptr(y1), B, tokens, d, C, L.dtype_code(dtype), L.stream()))
	assert relerr(y0, xr[:, 0] @ w0r.t() + b0r) < TOL[dtype]
	assert relerr(y1, xr[:, 1] @ w1r.t() + b1r) < TOL[dtype]


@pytest.mark.parametrize("B,h,Sq,Skv,p", [
	(2, 2, 64, 64, 0.0), (2, 2, 65, 65, 0.0), (2, 3, 80, 80, 0.0), (1, 2, 96, 96, 0.0), (2, 2, 97, 97, 0.0), (1, 4, 128, 128, 0.0), (2, 12, 197, 197, 0.0),
	(1, 2, 224, 224, 0.0), (1, 2, 256, 224, 0.0), (2, 2, 33, 17, 0.0), (2, 3, 256, 64, 0.0), (1, 2, 40, 1, 0.0), (3, 2, 197, 49, 0.0),
	(64, 12, 197, 197, 0.0), (70, 12, 65, 65, 0.0),  # more (batch, head) pairs than CUs: a workgroup walks several (768: three, 840: three or four)
	(2, 12, 197, 197, 0.3), (3, 2, 96, 80, 0.2), (70, 12, 65, 65, 0.3)])
def test_attention_backward_in_one_pass(gpu, B, h, Sq, Skv, p):
	"""The one-kernel backward (scores and their gradient formed once; key counts 64 ... 224 incl. the ragged 65 / 80 / 97 /
	197, fewer keys than queries, one key, many pairs per workgroup, dropout) against the CPU oracle at the kernel bar AND
	against the two-kernel form it replaces (same arithmetic per element, different summation orders: 3e-2 of the bf16
	result's largest value)."""
	from vit_amd import ops
	import vit_amd._lib as L
	dh, dtype = 64, torch.bfloat16
	d = h * dh
	seed = 0x0BAD_5EED + Sq
	q, qr = rnd((B * Sq, d), dtype, gpu, seed=1)
	k, kr = rnd((B * Skv, d), dtype, gpu, seed=2)
	v, vr = rnd((B * Skv, d), dtype, gpu, seed=3)
	do, dor = rnd((B * Sq, d), dtype, gpu, seed=4)
	drop = (p, seed) if p > 0 else None
	kw = dict(drop=drop) if drop else {}
	out, lse = ops.attention_fwd(q, k, v, B, h, Sq, Skv, dh, **kw)
	res = {}
	lib = L.load()
	was = lib.vitk_attention_bwd_one_pass(1)
	try:
		for mode in (1, 0):
			lib.vitk_attention_bwd_one_pass(mode)
			dq, dk, dv = torch.full_like(q, 7.0), torch.full_like(k, 7.0), torch.full_like(v, 7.0)
			ops.attention_bwd(do, q, k, v, out, lse, dq, dk, dv, B, h, Sq, Skv, dh, **kw)
			torch.cuda.synchronize()
			res[mode] = (dq, dk, dv)
	finally:
		lib.vitk_attention_bwd_one_pass(was)
	for a_, b_, name in zip(res[1], res[0], ("dq", "dk", "dv")):
		assert torch.isfinite(a_.float()).all(), name
		scale = float(b_.float().abs().max())
		assert float((a_.float() - b_.float()).abs().max()) <= 3e-2 * scale + 1e-6, name
	if B * h > 200:
		return  # (the oracle comparison of the big cases would take minutes on the CPU; the two-kernel form is oracle-checked elsewhere)
	q4 = O.split_heads(qr.reshape(B, Sq, d), h)
	k4 = O.split_heads(kr.reshape(B, Skv, d), h)
	v4 = O.split_heads(vr.reshape(B, Skv, d), h)
	do4 = O.split_heads(dor.reshape(B, Sq, d), h)
	if drop:
		keep = O.attn_dropout_keep(seed, p, B, h, Sq, Skv)
		o_ref, lse_ref = O.sdpa_dropout_fwd(q4, k4, v4, keep)
		refs = O.sdpa_dropout_bwd(do4, q4, k4, v4, o_ref, lse_ref, keep)
	else:
		o_ref, lse_ref = O.sdpa_fwd(q4, k4, v4)
		refs = O.sdpa_bwd(do4, q4, k4, v4, o_ref, lse_ref)
	dq, dk, dv = res[1]
	if Skv == 1:
		assert dq.abs().max().item() < 5e-2 and dk.abs().max().item() < 5e-2
	else:
		assert relerr(dq, O.merge_heads(refs[0]).reshape(B * Sq, d)) < TOL[dtype]
		assert relerr(dk, O.merge_heads(refs[1]).reshape(B * Skv, d)) < TOL[dtype]
	assert relerr(dv, O.merge_heads(refs[2]).reshape(B * Skv, d)) < TOL[dtype]
