"""Random attention problems (bf16) through vitk_attention_fwd / _bwd against the float64 oracle: python tools/diag/attn_fuzz.py [cases] [seed].
Covers every kernel family by head size and key count; prints the first case over the bf16 bar and exits 1."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from oracle import vit_oracle as O
from vit_amd import ops

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def relerr(a, b):
	a, b = a.detach().double().cpu(), b.detach().double().cpu()
	return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


worst = 0.0
for case in range(cases):
	dh = rng.choice([4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 64, 64])
	h = rng.randint(1, 6)
	B = rng.randint(1, 5)
	Sq = rng.choice([1, 5, 16, 17, 33, 64, 65, 100, 128, 197, 200, 256, 300, 520])
	Skv = rng.choice([2, 4, 13, 16, 17, 64, 65, 96, 129, 197, 224, 250, 256, 257, 300])
	d = h * dh
	g = torch.Generator().manual_seed(case)
	mk = lambda rows: torch.randn(rows, d, generator=g).to(torch.bfloat16)
	qh, kh, vh, doh = mk(B * Sq), mk(B * Skv), mk(B * Skv), mk(B * Sq)
	q, k, v, do = (t.to(dev) for t in (qh, kh, vh, doh))
	out, lse = ops.attention_fwd(q, k, v, B, h, Sq, Skv, dh)
	dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
	ops.attention_bwd(do, q, k, v, out, lse, dq, dk, dv, B, h, Sq, Skv, dh)
	q4, k4, v4, do4 = (O.split_heads(t.double().reshape(B, -1, d), h) for t in (qh, kh, vh, doh))
	o_ref, lse_ref = O.sdpa_fwd(q4, k4, v4)
	dq_r, dk_r, dv_r = O.sdpa_bwd(do4, q4, k4, v4, o_ref, lse_ref)
	errs = dict(out=relerr(out, O.merge_heads(o_ref).reshape(B * Sq, d)), dq=relerr(dq, O.merge_heads(dq_r).reshape(B * Sq, d)),
		dk=relerr(dk, O.merge_heads(dk_r).reshape(B * Skv, d)), dv=relerr(dv, O.merge_heads(dv_r).reshape(B * Skv, d)))
	worst = max(worst, max(errs.values()))
	if max(errs.values()) > 2e-2 or any(e != e for e in errs.values()):
		print("OVER THE BAR", dict(case=case, B=B, h=h, Sq=Sq, Skv=Skv, dh=dh), errs, flush=True)
		sys.exit(1)
	if case % 20 == 19:
		print(case + 1, "cases ok, worst", round(worst, 5), flush=True)
print("all ok, worst", worst)
