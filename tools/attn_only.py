"""Runs the attention forward and backward at the ViT-B/16 batch-256 shape a few times (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

dev = torch.device("cuda:0")
B, S, d, h, dh = 256, 197, 768, 12, 64
M = B * S
qkv = torch.randn(M, 3 * d, device=dev).to(torch.bfloat16)
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
dout = torch.randn(M, d, device=dev).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
for _ in range(3):
	out, lse = ops.attention_fwd(q, k, v, B, h, S, S, dh)
	ops.attention_bwd(dout, q, k, v, out, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, h, S, S, dh)
torch.cuda.synchronize()
