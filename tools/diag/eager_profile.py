"""cProfile of the eager training step of a bench configuration (host-side cost per step: the launch-bound 32x32
configurations spend more time in Python than on the GPU).  python tools/diag/eager_profile.py cvt"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vit_amd import dp, ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cvt"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
if name == "cvt":
	from vit_amd.models import CvT
	c = bench.CVT
	model = CvT(bench.make_cvt_config()).to(dev)
else:
	from vit_amd.models import ViT
	c = bench.CONFIGS[name]
	model = ViT(bench.make_config(c)).to(dev)
trainer = dp.Trainer(model, lr=1e-4, weight_decay=0.05)
x = torch.randn(c["batch"], *c["chw"], device=dev)
y = torch.randint(0, c["out_d"], (c["batch"],), device=dev)


def step():
	trainer.zero_grad()
	with torch.autocast("cuda", dtype=torch.bfloat16):
		loss = ops.cross_entropy(model(x), y)
	loss.backward()
	trainer.step()


for _ in range(5):
	step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
	step()
torch.cuda.synchronize()
print(f"eager: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
	step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
