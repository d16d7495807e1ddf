"""Micro-benchmark of vitk_gemm on the small problems of the 32x32 configurations (classifier head of 10 classes, patch
embedding of 4x4 patches): microseconds per launch, back-to-back launches, next to torch.matmul on the same operands."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vit_amd._lib as L
from tools.bench_gemm import timeit
from vit_amd import ops

dev = torch.device("cuda:0")
bf = torch.bfloat16
lib = L.load()


def line(name, fn, ref):
	t = timeit(fn, iters=50) * 1e6
	k = lib.vitk_gemm_last_kernel().decode()
	r = timeit(ref, iters=50) * 1e6
	print(f"{name:<44} {t:7.1f} us  {k:<22} torch {r:7.1f} us", flush=True)


for B, C, d, S in ((512, 10, 192, 65), (256, 10, 384, 1), (256, 1000, 768, 197)):
	x = torch.randn(B * S, d, device=dev).to(bf)
	rows0 = x[::S]
	w = (torch.randn(C, d, device=dev) * 0.05).to(bf)
	b = torch.randn(C, device=dev)
	dy = torch.randn(B, C, device=dev).to(bf)
	line(f"head fwd   [{B}x{d}] @ [{C}x{d}]^T", lambda: ops.linear_fwd(rows0, w, b), lambda: torch.matmul(rows0, w.t()))
	dxb = torch.zeros(B * S, d, device=dev, dtype=bf)
	line(f"head dgrad [{B}x{C}] @ [{C}x{d}]", lambda: ops.linear_dgrad(dy, w, out=dxb[::S]), lambda: torch.matmul(dy, w))
	line(f"head wgrad [{B}x{C}]^T @ [{B}x{d}] (+bias grad)", lambda: ops.linear_wgrad(dy, rows0, bias_grad=True), lambda: torch.matmul(dy.t(), rows0))

# patch embedding of ViT-Tiny/4: 512 images of 3x32x32, 8x8 patches of 4x4 pixels, K = 48
img = torch.randn(512, 3, 32, 32, device=dev).to(bf)
w = (torch.randn(192, 48, device=dev) * 0.05).to(bf)
bias = torch.randn(192, device=dev)
cls = torch.randn(1, 1, 192, device=dev)
pos = torch.randn(1, 65, 192, device=dev)
with torch.autocast("cuda", dtype=bf):
	line("patch embed fwd 512 x 64 patches of 48 -> 192", lambda: ops.PatchEmbedFn.apply(img, w, bias, cls, pos, 8),
		lambda: torch.nn.functional.conv2d(img, w.reshape(192, 3, 4, 4), stride=4))
