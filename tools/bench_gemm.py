"""Micro-benchmark of vitk_gemm on the ViT-B/16 shapes (random bf16 data): TFLOP/s per layout."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
import vit_amd._lib as L

dev = torch.device("cuda:0")
M = 256 * 197
shapes = [("qkv", M, 2304, 768), ("proj", M, 768, 768), ("fc_in", M, 3072, 768), ("fc_out", M, 768, 3072)]


def timeit(fn, iters=20, reps=3):
	"""best of `reps` averages over `iters` launches"""
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


for name, m, n, k in (shapes if __name__ == "__main__" else []):
	x = torch.randn(m, k, device=dev).to(torch.bfloat16)
	w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
	dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
	bias = torch.randn(n, device=dev)
	y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
	t = timeit(lambda: ops.gemm(x, w, y, m, n, k, k, k, n))
	t2 = timeit(lambda: ops.linear_fwd(x, w, bias, act=L.ACT_GELU, save_pre=True))
	dx = torch.empty(m, k, device=dev, dtype=torch.bfloat16)
	t3 = timeit(lambda: ops.linear_dgrad(dy, w, out=dx))
	t4 = timeit(lambda: ops.linear_wgrad(dy, x))
	fl = 2.0 * m * n * k / 1e12
	ref = timeit(lambda: torch.matmul(x, w.t()))
	print(f"{name:7s} M={m} N={n} K={k}: fwd {fl/t:7.1f}  fwd+bias+gelu+pre {fl/t2:7.1f}  dgrad {fl/t3:7.1f}  wgrad {fl/t4:7.1f}  TFLOP/s   (torch.matmul {fl/ref:7.1f})", flush=True)
