"""Runs the attention forward and backward at the ViT-B/16 batch-256 shape a few times (for rocprofv3 counter passes)
and prints their times.  Optional argument: dropout probability on the attention probabilities."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_amd import ops

p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
drop = (p, 12345)
dev = torch.device("cuda:0")
B, S, d, h, dh = 256, 197, 768, 12, 64
M = B * S
qkv = torch.randn(M, 3 * d, device=dev).to(torch.bfloat16)
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
dout = torch.randn(M, d, device=dev).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf, tb = [], []
for _ in range(8):
	ev[0].record()
	out, lse = ops.attention_fwd(q, k, v, B, h, S, S, dh, drop=drop)
	ev[1].record()
	ops.attention_bwd(dout, q, k, v, out, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, h, S, S, dh, drop=drop)
	ev[2].record()
	torch.cuda.synchronize()
	tf.append(ev[0].elapsed_time(ev[1]))
	tb.append(ev[1].elapsed_time(ev[2]))
print(f"dropout {p}: forward {1e3 * min(tf):.1f} us, backward (dQ + dK/dV) {1e3 * min(tb):.1f} us")
