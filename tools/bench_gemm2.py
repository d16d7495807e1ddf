import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
	for _ in range(3): fn()
	torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(iters): fn()
	e1.record(); torch.cuda.synchronize()
	return e0.elapsed_time(e1) / iters * 1e-3
for (m, n, k) in [(50432, 3072, 768), (50432, 768, 3072), (8192, 8192, 8192)]:
	x = torch.randn(m, k, device=dev).to(torch.bfloat16)
	w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
	y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
	t = timeit(lambda: ops.gemm(x, w, y, m, n, k, k, k, n))
	print(f"dbg={os.environ.get('VITK_DBG','0')} M={m} N={n} K={k}: {2.0*m*n*k/t/1e12:7.1f} TFLOP/s  {t*1e6:8.1f} us", flush=True)
