"""Attention backward at the ViT-B/16 batch-256 shape (3072 (batch, head) pairs, 197 tokens, head size 64, bf16): the
one-kernel form against the two-kernel form, time per layer and the fraction of the HBM roofline (algorithmic bytes: q, k,
v, dO, O read and dq, dk, dv written once = 8 images of 77.5 MB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
import vit_amd._lib as L

dev = torch.device("cuda:0")
B, S, d, h, dh = 256, 197, 768, 12, 64
M = B * S
bf = torch.bfloat16
torch.manual_seed(0)
qkv = torch.randn(M, 3 * d, device=dev).to(bf)
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
out, lse = ops.attention_fwd(q, k, v, B, h, S, S, dh)
dout = torch.randn(M, d, device=dev).to(bf)
dqkv = torch.empty_like(qkv)
lib = L.load()


def run():
	ops.attention_bwd(dout, q, k, v, out, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, h, S, S, dh)


def timeit(iters=30, reps=3):
	for _ in range(3):
		run()
	best = 1e9
	for _ in range(reps):
		torch.cuda.synchronize()
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record()
		for _ in range(iters):
			run()
		e1.record()
		torch.cuda.synchronize()
		best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
	return best


nbytes = 8 * M * d * 2
for mode, name in ((0, "two kernels (dQ, then dK/dV)"), (1, "one pass"), (0, "two kernels (dQ, then dK/dV)"), (1, "one pass")):
	lib.vitk_attention_bwd_one_pass(mode)
	t = timeit()
	print(f"{name:32s} {t * 1e6:7.1f} us per layer   {nbytes / t / 1e12:5.2f} TB/s = {nbytes / t / 8e12:4.2f} of the 8 TB/s HBM roofline", flush=True)
