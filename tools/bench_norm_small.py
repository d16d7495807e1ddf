"""LayerNorm / RMSNorm forward and backward on the narrow rows of the 32x32 configurations (bf16): microseconds and GB/s of
algorithmic bytes per launch.  Run twice, once with VITK_NO_NORM_SMALL_ROWS=1, to compare one row per wavefront with
several rows per wavefront."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_gemm import timeit
from vit_amd import ops

dev = torch.device("cuda:0")
SHAPES = ((33280, 192), (16384, 64), (4096, 192), (1024, 384), (100352, 96), (25088, 384), (33280, 512))
if len(sys.argv) > 2:  # one shape only (under a kernel trace: one kernel name, one shape)
	SHAPES = ((int(sys.argv[1]), int(sys.argv[2])),)
for kind in ("layernorm", "rmsnorm"):
	for M, d in SHAPES:
		w = torch.ones(d, device=dev)
		b = torch.zeros(d, device=dev) if kind == "layernorm" else None
		x = torch.randn(M, d, device=dev).to(torch.bfloat16)
		dy = torch.randn(M, d, device=dev).to(torch.bfloat16)
		res = torch.randn(M, d, device=dev).to(torch.bfloat16)
		y, _, mean, rstd = ops.norm_fwd(x, w, b, kind, 1e-5)
		tf = timeit(lambda: ops.norm_fwd(x, w, b, kind, 1e-5), iters=50) * 1e6
		tb = timeit(lambda: ops.norm_bwd(dy, x, w, mean, rstd, kind, dres=res), iters=50) * 1e6
		print(f"{kind:9s} rows {M:6d} d {d:4d}: fwd {tf:6.1f} us {2 * M * d * 2 / tf / 1e3:6.0f} GB/s   bwd(+dres) {tb:6.1f} us {4 * M * d * 2 / tb / 1e3:6.0f} GB/s", flush=True)
