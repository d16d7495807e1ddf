"""Diagnostic: flat gradients of a small ViT step with weight gradients on the second stream vs on the main stream,
and main stream vs main stream (run-to-run determinism), per parameter."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_gpu_dp import _model, CFG
from vit_amd import dp, ops
gpu = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
x = torch.randn(32, 3, 16, 16, generator=g).to(gpu)
y = torch.randint(0, 10, (32,), generator=g).to(gpu)
ops.WGRAD_OVERLAP_MIN_FLOPS = 0.0
res = []
for mode in (False, False, True, True):
	ops.WGRAD_OVERLAP = mode
	ops.clear_grad_sinks()
	model = _model(gpu)
	tr = dp.Trainer(model, lr=1e-2, weight_decay=0.05)
	tr.zero_grad()
	with torch.autocast("cuda", dtype=torch.bfloat16):
		ops.cross_entropy(model(x), y).backward()
	torch.cuda.synchronize()
	res.append((tr.flat.grad.clone(), tr.flat.names, tr.flat.offsets, [p.numel() for p in tr.flat.params]))
	tr.close()
for a, b, tag in ((0, 1, "main vs main"), (0, 2, "main vs second"), (2, 3, "second vs second")):
	ga, gb = res[a][0], res[b][0]
	print(tag, "equal" if torch.equal(ga, gb) else "DIFFER")
	for n, o, k in zip(res[a][1], res[a][2], res[a][3]):
		d = (ga[o:o + k] - gb[o:o + k]).abs().max().item()
		if d > 0:
			print("   ", n, d, ga[o:o + k].abs().max().item())
