"""LayerNorm backward time against the number of rows (d = 768, bf16): separates the fixed cost of a launch (prologue,
the dgamma / dbeta reduction) from the per-row streaming cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

dev = torch.device("cuda:0")
d = 768
w = torch.ones(d, device=dev)
b = torch.zeros(d, device=dev)
for M in (50432, 25216, 12608, 6304, 3152):
	x = torch.randn(M, d, device=dev).to(torch.bfloat16)
	dy = torch.randn(M, d, device=dev).to(torch.bfloat16)
	res = torch.randn(M, d, device=dev).to(torch.bfloat16)
	y, _, mean, rstd = ops.norm_fwd(x, w, b, "layernorm", 1e-5)
	ts = []
	for _ in range(20):
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record()
		ops.norm_bwd(dy, x, w, mean, rstd, "layernorm", dres=res)
		e1.record()
		torch.cuda.synchronize()
		ts.append(e0.elapsed_time(e1) * 1e3)
	t = sorted(ts)[len(ts) // 2]
	print(f"rows {M:6d}: {t:6.1f} us  {4 * M * d * 2 / t / 1e3:7.0f} GB/s")
