"""Runs each ViT-B/16 GEMM flavour a few times (for rocprofv3 --pmc passes): fwd, fwd+gelu, dgrad, dgrad+gelu', wgrad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops
import vit_amd._lib as L

dev = torch.device("cuda:0")
m, n, k = 256 * 197, 3072, 768
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
bias = torch.randn(n, device=dev)
y = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
dx = torch.empty(m, k, device=dev, dtype=torch.bfloat16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
	ops.gemm(x, w, y, m, n, k, k, k, n)
	_, pre = ops.linear_fwd(x, w, bias, act=L.ACT_GELU, save_pre=True)
	ops.linear_dgrad(dy, w, out=dx)
	ops.linear_dgrad(x, w.t().contiguous(), act=L.ACT_GELU, pre=pre)
	ops.linear_wgrad(dy, x)
torch.cuda.synchronize()
