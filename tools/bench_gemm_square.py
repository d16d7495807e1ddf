"""vitk_gemm (K-major x K-major, bf16 out) against torch.matmul on square shapes, random data: TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
from bench_gemm import timeit

dev = torch.device("cuda:0")
for n in (4096, 8192):
	for k in (n, 768):
		x = torch.randn(n, k, device=dev).to(torch.bfloat16)
		w = torch.randn(n, k, device=dev).to(torch.bfloat16)
		y = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
		t = timeit(lambda: ops.gemm(x, w, y, n, n, k, k, k, n))
		ref = timeit(lambda: torch.matmul(x, w.t()))
		fl = 2.0 * n * n * k / 1e12
		print(f"M=N={n} K={k}: vitk {fl/t:7.1f}  torch.matmul {fl/ref:7.1f} TFLOP/s", flush=True)
