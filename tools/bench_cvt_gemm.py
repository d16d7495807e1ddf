"""Micro-benchmark of the forward / input-gradient GEMMs of the CvT stages (config 3: 65536 x 64, 16384 x 192, 4096 x 384 tokens x
width at batch 256): microseconds per launch, back to back, next to torch.matmul on the same operands."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vit_amd._lib as L
from tools.bench_gemm import timeit
from vit_amd import ops

dev = torch.device("cuda:0")
bf = torch.bfloat16
lib = L.load()


def line(name, fn, ref, flops):
	t = timeit(fn, iters=50) * 1e6
	k = lib.vitk_gemm_last_kernel().decode()
	r = timeit(ref, iters=50) * 1e6
	print(f"{name:<52} {t:7.1f} us {flops / t / 1e6:7.1f} TFLOP/s  {k:<22} torch {r:7.1f} us", flush=True)


for T, C in ((65536, 64), (16384, 192), (4096, 384)):
	x = torch.randn(T, C, device=dev).to(bf)
	h = torch.randn(T, 4 * C, device=dev).to(bf)
	q3 = torch.randn(T, 3 * C, device=dev).to(bf)
	w3 = (torch.randn(3 * C, C, device=dev) * 0.05).to(bf)
	wo = (torch.randn(C, C, device=dev) * 0.05).to(bf)
	w1 = (torch.randn(4 * C, C, device=dev) * 0.05).to(bf)
	w2 = (torch.randn(C, 4 * C, device=dev) * 0.05).to(bf)
	b3, bo, b1, b2 = (torch.randn(n, device=dev) for n in (3 * C, C, 4 * C, C))
	print(f"-- {T} tokens x {C}")
	line("q|k|v fwd  (+bias)", lambda: ops.linear_fwd(x, w3, b3), lambda: torch.matmul(x, w3.t()), 2.0 * T * 3 * C * C)
	line("proj fwd   (+bias, +residual)", lambda: ops.linear_fwd(x, wo, bo, residual=x), lambda: torch.matmul(x, wo.t()), 2.0 * T * C * C)
	line("fc_in fwd  (+bias, gelu, pre)", lambda: ops.linear_fwd(x, w1, b1, act=L.ACT_GELU, save_pre=True), lambda: torch.matmul(x, w1.t()), 2.0 * T * 4 * C * C)
	line("fc_out fwd (+bias, +residual)", lambda: ops.linear_fwd(h, w2, b2, residual=x), lambda: torch.matmul(h, w2.t()), 2.0 * T * 4 * C * C)
	dx = torch.empty(T, C, device=dev, dtype=bf)
	line("q|k|v dgrad (stacked weight, MN-major)", lambda: ops.gemm(q3, w3, dx, T, C, 3 * C, 3 * C, C, C, b_layout=L.MN_MAJOR), lambda: torch.matmul(q3, w3), 2.0 * T * 3 * C * C)
	line("proj dgrad", lambda: ops.linear_dgrad(x, wo), lambda: torch.matmul(x, wo), 2.0 * T * C * C)
	line("fc_out dgrad (gelu')", lambda: ops.linear_dgrad(x, w2, act=L.ACT_GELU, pre=h), lambda: torch.matmul(x, w2), 2.0 * T * 4 * C * C)
	line("fc_in dgrad", lambda: ops.linear_dgrad(h, w1), lambda: torch.matmul(h, w1), 2.0 * T * 4 * C * C)
