"""The fc_in weight-gradient GEMM of ViT-B/16 (batch 256) a few times, for rocprofv3 counter passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

dev = torch.device("cuda:0")
m, n, k = 256 * 197, 3072, 768
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
for _ in range(3):
	ops.linear_wgrad(dy, x)
torch.cuda.synchronize()
