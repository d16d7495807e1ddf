"""Every vitk_gemm launch of one training step of a bench configuration: shape, layouts, epilogue, the kernel family that
ran and its duration (HIP events around the launch), grouped by identical launches.
python tools/diag/gemm_shapes.py vit_tiny"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vit_amd import dp, ops  # noqa: E402
from vit_amd import _lib as L  # noqa: E402
from vit_amd.models import ViT  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "vit_tiny"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
if name == "cvt":
	from vit_amd.models import CvT
	c = bench.CVT
	model = CvT(bench.make_cvt_config()).to(dev)
else:
	c = bench.CONFIGS[name]
	model = ViT(bench.make_config(c)).to(dev)
trainer = dp.Trainer(model, lr=1e-4, weight_decay=0.05)
x = torch.randn(c["batch"], *c["chw"], device=dev)
y = torch.randint(0, c["out_d"], (c["batch"],), device=dev)
records = []
real_gemm = ops.gemm
lib = L.load()


def spy(A, B, C, M, N, K, lda, ldb, ldc, a_layout=L.K_MAJOR, b_layout=L.K_MAJOR, **kw):
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	out = real_gemm(A, B, C, M, N, K, lda, ldb, ldc, a_layout=a_layout, b_layout=b_layout, **kw)
	e1.record()
	key = (M, N, K, "MN" if a_layout == L.MN_MAJOR else "K", "MN" if b_layout == L.MN_MAJOR else "K", str(C.dtype).replace("torch.", ""),
		kw.get("epilogue", 0), "patch" if kw.get("patch") else "", lib.vitk_gemm_last_kernel().decode())
	records.append((key, e0, e1))
	return out


def step():
	trainer.zero_grad()
	with torch.autocast("cuda", dtype=torch.bfloat16):
		loss = ops.cross_entropy(model(x), y)
	loss.backward()
	trainer.step()


for _ in range(3):
	step()
ops.gemm = spy
for _ in range(5):
	step()
torch.cuda.synchronize()
agg = {}
for key, e0, e1 in records:
	a = agg.setdefault(key, [0, 0.0])
	a[0] += 1
	a[1] += e0.elapsed_time(e1)
print(f"{'M':>7} {'N':>6} {'K':>7} A  B  out      epi patch  kernel                     n/step  us each  us/step")
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
	M, N, K, la, lb, dt, epi, patch, kern = key
	print(f"{M:>7} {N:>6} {K:>7} {la:<2} {lb:<2} {dt:<8} {epi:>3} {patch:<6} {kern:<26} {n / 5:>6.1f} {1e3 * ms / n:>8.1f} {1e3 * ms / 5:>8.1f}")
