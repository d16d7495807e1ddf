"""The attention backward that takes its scratch from a fixed array inside the library (for callers that cannot allocate)
against the entry point that takes a caller buffer: bit-identical dq / dk / dv, on two calls in a row, the second with a
smaller problem (stale contents of the shared array must not leak), and a clear error beyond the array's size."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(gpu, B, H, S, Skv, dh, seed, drop):
	from vit_amd import _lib as L
	lib = L.load()
	g = torch.Generator().manual_seed(seed)
	d = H * dh
	mk = lambda rows: (torch.randn(B, rows, d, generator=g) * 0.5).to(torch.bfloat16).to(gpu)
	q, k, v, dout = mk(S), mk(Skv), mk(Skv), mk(S)
	out = torch.empty_like(q)
	lse = torch.empty(B, H, S, dtype=torch.float32, device=gpu)
	st = L.stream()
	L.check(lib.vitk_attention_fwd_dropout(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), B, H, S, Skv, dh, d, d, d, d, 0.0, drop, 77, L.BF16, st))
	res = []
	for which in range(2):
		dq, dk, dv = torch.full_like(q, 7.0), torch.full_like(k, 7.0), torch.full_like(v, 7.0)
		if which == 0:
			scratch = torch.empty(2 * B * H * S, dtype=torch.float32, device=gpu)
			L.check(lib.vitk_attention_bwd_dropout(L.ptr(dout), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), L.ptr(dq), L.ptr(dk), L.ptr(dv),
				B, H, S, Skv, dh, d, d, d, d, d, d, d, 0.0, drop, 77, L.ptr(scratch), L.BF16, st))
		else:
			L.check(lib.vitk_attention_bwd_noalloc(L.ptr(dout), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), L.ptr(dq), L.ptr(dk), L.ptr(dv),
				B, H, S, Skv, dh, d, d, d, d, d, d, d, 0.0, drop, 77, L.BF16, st))
		torch.cuda.synchronize()
		res.append((dq, dk, dv))
	for a, b, name in zip(res[0], res[1], ("dq", "dk", "dv")):
		assert torch.isfinite(b.float()).all(), name
		assert torch.equal(a, b), f"{name} differs between the caller-scratch and the fixed-scratch entry points"


def test_noalloc_attention_backward_matches_twice_in_a_row(gpu):
	_run(gpu, 8, 12, 197, 197, 64, 1, 0.0)  # the ViT-B/16 head geometry (MFMA kernels)
	_run(gpu, 2, 3, 65, 65, 64, 2, 0.0)     # smaller, right after: the array still holds the first call's statistics
	_run(gpu, 3, 6, 49, 16, 64, 3, 0.25)    # fewer keys than queries, with dropout on the probabilities
	_run(gpu, 2, 4, 40, 40, 24, 4, 0.0)     # a head size the generic kernels take (they ignore the scratch)


def test_noalloc_attention_backward_rejects_what_does_not_fit(gpu):
	from vit_amd import _lib as L
	lib = L.load()
	B, H, S, dh = 1024, 12, 197, 64  # 2 * B * H * S = 4.8 M floats > 2 Mi
	t = torch.zeros(8, dtype=torch.bfloat16, device=gpu)
	f = torch.zeros(8, dtype=torch.float32, device=gpu)
	rc = lib.vitk_attention_bwd_noalloc(L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(f), L.ptr(t), L.ptr(t), L.ptr(t),
		B, H, S, S, dh, H * dh, H * dh, H * dh, H * dh, H * dh, H * dh, H * dh, 0.0, 0.0, 0, L.BF16, L.stream())
	assert rc == -6 and b"scratch" in lib.vitk_strerror(rc)
	# sizes whose product 2 * B * H * S overflows 64 bits are rejected as well, not wrapped into something that fits
	big = (1 << 31) - 1
	for Bx, Hx, Sx in ((big, big, big), (1 << 30, 1 << 30, 8), (big, 1, 1), (1, big, big)):
		rc = lib.vitk_attention_bwd_noalloc(L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(t), L.ptr(f), L.ptr(t), L.ptr(t), L.ptr(t),
			Bx, Hx, Sx, Sx, dh, Hx * dh, Hx * dh, Hx * dh, Hx * dh, Hx * dh, Hx * dh, Hx * dh, 0.0, 0.0, 0, L.BF16, L.stream())
		assert rc == -6, (Bx, Hx, Sx, rc)
