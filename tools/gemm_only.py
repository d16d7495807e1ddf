"""One large K-major x K-major bf16 GEMM a few times (for rocprofv3 counter passes on the persistent kernel's main loop)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x = torch.randn(n, n, device=dev).to(torch.bfloat16)
w = torch.randn(n, n, device=dev).to(torch.bfloat16)
y = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
for _ in range(3):
	ops.gemm(x, w, y, n, n, n, n, n, n)
torch.cuda.synchronize()
