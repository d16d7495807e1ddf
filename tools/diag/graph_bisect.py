"""Bisects what in bench.py's sequence around a captured CvT step changes its training trajectory."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from vit_amd import dp, ops
from vit_amd.models import CvT
V = os.environ.get("V", "")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(0)
model = CvT(bench.make_cvt_config(0.0)).to(dev)
trainer = dp.Trainer(model, lr=1e-4, weight_decay=0.05)
g = torch.Generator(device="cpu").manual_seed(1234)
x = torch.randn(256, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (256,), generator=g).to(dev)


def loss_of(xi, yi):
	with torch.autocast("cuda", dtype=torch.bfloat16):
		return ops.cross_entropy(model(xi), yi)


graphed = dp.GraphedStep(trainer, loss_of, [x, y], warmup=3)
first = None
for i in range(3):
	l0 = graphed(x, y)
	if first is None and "noclone" not in V:
		first = l0.detach().clone()
if "sleep" in V:
	time.sleep(0.2)
elif "streamsync" in V:
	torch.cuda.current_stream().synchronize()
elif "item" in V:
	l0.item()
elif "nosync" not in V:
	torch.cuda.synchronize()
for i in range(10):
	loss = graphed(x, y)
torch.cuda.synchronize()
if "tensor" in V:
	t = torch.tensor([1.0], dtype=torch.float64, device=dev)
print(V or "bench-like", "final", round(loss.item(), 4), "first", None if first is None else round(first.item(), 4))
