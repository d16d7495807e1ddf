"""Per-step losses of the CvT bench configuration run eagerly and as a captured HIP graph (same data, same initial weights)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from vit_amd import dp, ops
from vit_amd.models import CvT
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
x = torch.randn(256, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (256,), generator=g).to(dev)
out = {}
for mode in (os.environ.get("MODES", "eager,graph").split(",")):
	ops.clear_grad_sinks()
	torch.manual_seed(0)
	model = CvT(bench.make_cvt_config(0.0)).to(dev)
	tr = dp.Trainer(model, lr=1e-4, weight_decay=0.05)

	def loss_of(xi, yi):
		with torch.autocast("cuda", dtype=torch.bfloat16):
			return ops.cross_entropy(model(xi), yi)
	losses = []
	if mode == "eager":
		for _ in range(13):
			tr.zero_grad()
			l = loss_of(x, y)
			l.backward()
			tr.step()
			losses.append(float(l.detach()))
	else:
		gs = dp.GraphedStep(tr, loss_of, [x, y])
		nosync = os.environ.get("NOSYNC", "0") == "1"
		keep = []
		for _ in range(13):
			l = gs(x, y)
			if os.environ.get("CLONE", "0") == "2" or (os.environ.get("CLONE", "0") == "1" and not keep):
				keep.append(l.detach().clone())  # an allocation behind the capture, as a caller that keeps the first loss
			if not nosync:
				losses.append(float(l))
		if nosync:
			torch.cuda.synchronize()
			losses.append(float(l))
	bn = [b.float().abs().sum().item() for n, b in model.named_buffers()][:4]
	print(mode, [round(v, 4) for v in losses], "buffers", [round(v, 3) for v in bn], flush=True)
	tr.close()
