"""Micro-benchmark of the non-GEMM kernels at the ViT-B/16 batch-256 shapes (bf16): time and effective HBM GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
import vit_amd._lib as L

dev = torch.device("cuda:0")
B, S, d, h, dh = 256, 197, 768, 12, 64
M = B * S
bf = torch.bfloat16


def timeit(fn, iters=20, reps=3):
	for _ in range(3):
		fn()
	best = 1e9
	for _ in range(reps):
		torch.cuda.synchronize()
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record()
		for _ in range(iters):
			fn()
		e1.record()
		torch.cuda.synchronize()
		best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
	return best


def report(name, t, nbytes=None, flops=None):
	s = f"{name:34s} {t*1e6:8.1f} us"
	if nbytes:
		s += f"  {nbytes/t/1e9:8.0f} GB/s"
	if flops:
		s += f"  {flops/t/1e12:8.1f} TFLOP/s"
	print(s, flush=True)


x = torch.randn(M, d, device=dev).to(bf)
dy = torch.randn(M, d, device=dev).to(bf)
res = torch.randn(M, d, device=dev).to(bf)
w = torch.randn(d, device=dev)
b = torch.randn(d, device=dev)
y, s_, mean, rstd = ops.norm_fwd(x, w, b, "layernorm", 1e-5, res=res)
report("layernorm fwd (+residual)", timeit(lambda: ops.norm_fwd(x, w, b, "layernorm", 1e-5, res=res)), 4 * M * d * 2)
report("layernorm bwd (+dres)", timeit(lambda: ops.norm_bwd(dy, s_, w, mean, rstd, "layernorm", dres=res)), 4 * M * d * 2)
report("layernorm bwd", timeit(lambda: ops.norm_bwd(dy, s_, w, mean, rstd, "layernorm")), 3 * M * d * 2)
for n in (768, 2304, 3072):
	g = torch.randn(M, n, device=dev).to(bf)
	report(f"colsum N={n}", timeit(lambda: ops.colsum(g)), M * n * 2)
qkv = torch.randn(M, 3 * d, device=dev).to(bf)
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
out, lse = ops.attention_fwd(q, k, v, B, h, S, S, dh)
fl = 4.0 * B * h * S * S * dh
report("attention fwd", timeit(lambda: ops.attention_fwd(q, k, v, B, h, S, S, dh)), flops=fl)
dout = torch.randn(M, d, device=dev).to(bf)
dqkv = torch.empty_like(qkv)
report("attention bwd", timeit(lambda: ops.attention_bwd(dout, q, k, v, out, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, h, S, S, dh)), flops=2.5 * fl)
