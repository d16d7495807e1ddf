import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
from vit_amd import ops
from bench_gemm import timeit
dev = torch.device("cuda:0")
for m, n, k in ((6912, 768, 3072), (6912, 768, 2304), (8192, 8192, 8192), (50432, 768, 768)):
	x = torch.randn(m, k, device=dev).to(torch.bfloat16)
	w = torch.randn(n, k, device=dev).to(torch.bfloat16)
	y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
	t = timeit(lambda: ops.gemm(x, w, y, m, n, k, k, k, n))
	print(f"M={m} N={n} K={k}: {t*1e6:7.1f} us {2.0*m*n*k/t/1e12:7.1f} TFLOP/s", flush=True)
