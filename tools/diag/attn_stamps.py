"""Cycle stamps of the one-pass attention backward (a library built with -DVITK_ATTN_STAMPS, VITK_LIB=diag_bin/libvitk_stamps.so):
cycles per query step in each segment of the step, per wave, averaged over the first 64 workgroups, at the ViT-B/16 shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vit_amd import ops
import vit_amd._lib as L

dev = torch.device("cuda:0")
B, S, d, h, dh = 256, 197, 768, 12, 64
M = B * S
bf = torch.bfloat16
torch.manual_seed(0)
qkv = torch.randn(M, 3 * d, device=dev).to(bf)
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
out, lse = ops.attention_fwd(q, k, v, B, h, S, S, dh)
dout = torch.randn(M, d, device=dev).to(bf)
dqkv = torch.empty_like(qkv)
lib = L.load()
scratch = torch.zeros(2 * B * h * S, dtype=torch.float32, device=dev)
for _ in range(3):
	L.check(lib.vitk_attention_bwd_dropout(L.ptr(dout), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), L.ptr(dqkv[:, :d]), L.ptr(dqkv[:, d:2 * d]), L.ptr(dqkv[:, 2 * d:]),
		B, h, S, S, dh, 3 * d, 3 * d, 3 * d, d, 3 * d, 3 * d, 3 * d, 0.0, 0.0, 0, L.ptr(scratch), L.BF16, L.stream()))
torch.cuda.synchronize()
st = scratch.view(torch.int32)[4096:4096 + 64 * 8 * 16].reshape(64, 8, 16).cpu().double()
names = ["pair start / loop", "row fragments + delta", "tile 0: S, dP, softmax, dS write", "tile 1", "dV / dK products", "dK / dV stores", "wait: own requests", "wait: barrier", "request issue", "dQ tile"]
steps = st[:, :, 13]
print("steps per workgroup:", steps[0, 0].item(), " total cycles per workgroup (wave 0):", st[:, 0, 12].mean().item())
print(f"{'segment':36s}" + "".join(f"  wave{w}" for w in range(8)))
tot = torch.zeros(8, dtype=torch.float64)
for i, n in enumerate(names):
	per = (st[:, :, i] / steps).mean(0)
	tot += per
	print(f"{n:36s}" + "".join(f"{per[w].item():7.0f}" for w in range(8)))
print(f"{'sum per step':36s}" + "".join(f"{tot[w].item():7.0f}" for w in range(8)))
