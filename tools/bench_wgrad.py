import sys, os
sys.path.insert(0, "/root/repo")
import torch
from vit_amd import ops
dev = torch.device("cuda:0")
M = 256 * 197
def timeit(fn, iters=10):
	for _ in range(3): fn()
	torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(iters): fn()
	e1.record(); torch.cuda.synchronize()
	return e0.elapsed_time(e1) / iters * 1e-3
for name, n, k in [("qkv", 2304, 768), ("proj", 768, 768), ("fc_in", 3072, 768), ("fc_out", 768, 3072)]:
	x = torch.randn(M, k, device=dev).to(torch.bfloat16)
	dy = torch.randn(M, n, device=dev).to(torch.bfloat16)
	t = timeit(lambda: ops.linear_wgrad(dy, x))
	print(f"{name:7s} wgrad {t*1e6:8.1f} us  {2.0*M*n*k/t/1e12:7.1f} TFLOP/s", flush=True)
