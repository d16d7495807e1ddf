"""The frozen ResNet-20 teacher on the GPU in the forms the distillation benchmark could run it: float32 / bf16 autocast,
BatchNorms kept / folded, NCHW / channels-last: error against the reference teacher's recorded logits and time per
forward at batch 512."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, relerr, sub  # noqa: E402
from vit_amd.models.teacher import frozen_teacher  # noqa: E402

dev = torch.device("cuda:0")
g, meta = load_golden("resnet20_teacher")
x8 = g["x"].to(dev)
xb = torch.randn(512, 3, 32, 32, device=dev)
for fold in (False, True):
	for cl in (False, True):
		for bf in (False, True):
			t = frozen_teacher(20, 10, sub(g, "sd"), fold_bn=fold).to(dev)
			fmt = torch.channels_last if cl else torch.contiguous_format
			t = t.to(memory_format=fmt)
			with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf):
				err = relerr(t(x8.contiguous(memory_format=fmt)), g["logits"])
				xi = xb.contiguous(memory_format=fmt)
				for _ in range(3):
					t(xi)
				torch.cuda.synchronize()
				t0 = time.perf_counter()
				for _ in range(10):
					t(xi)
				torch.cuda.synchronize()
				ms = (time.perf_counter() - t0) * 100
			print(f"fold={fold} channels_last={cl} bf16={bf}: err {err:.2e}  {ms:.3f} ms per forward at batch 512", flush=True)
