"""Random shapes / layouts / epilogues through vitk_gemm and vitk_gemm_group against float64 (small-integer operands, so the float32
results must be exact): python tools/diag/gemm_fuzz.py [cases] [seed] [big].  Prints the first mismatch and exits 1."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import vit_amd._lib as L
from vit_amd import ops

dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
big = len(sys.argv) > 3 and sys.argv[3] == "big"
lib = L.load()
seen = {}


def operand(rows, cols, mn_major, g):
	t = torch.randint(-3, 4, (rows, cols), generator=g).to(torch.bfloat16)
	return t, (t.t().contiguous() if mn_major else t).to(dev)


for case in range(cases):
	if big:  # whole-tile shapes of the deep-pipeline kernels and their neighbours
		M = rng.choice([2048, 4096, 8192, 12800, 12800 + 128, 25600, 50432])
		N = rng.choice([256, 320, 512, 576, 768, 1024, 2304])
		K = rng.choice([128, 192, 256, 768, 1536, 3072])
	else:
		M = rng.choice([8, 16, 64, 72, 128, 136, 200, 256, 264, 384, 520, 1000, 1024, 2048, 4096, 5000 // 8 * 8, 16384])
		N = rng.choice([8, 16, 64, 72, 128, 192, 264, 384, 576, 768, 1152, 1536])
		K = rng.choice([64, 128, 192, 256, 320, 384, 448, 768, 1024, 1536, 2304, 4096, 8192, 33280])
	if rng.random() < 0.5 and big:
		M, K = K, M  # (a weight-gradient shape: few output rows over a long K)
	if M * N * K > (4e11 if big else 4e10):
		continue
	a_mn, b_mn = rng.random() < 0.5, rng.random() < 0.5
	out_f32 = rng.random() < 0.5 or a_mn
	g = torch.Generator().manual_seed(case)
	a, A = operand(M, K, a_mn, g)
	b, B = operand(N, K, b_mn, g)
	ref = a.double() @ b.double().t()
	acc = out_f32 and rng.random() < 0.5
	colsum = a_mn and out_f32 and rng.random() < 0.5
	ws = rng.random() < 0.8
	c = torch.full((M, N), 2.0 if acc else float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
	cs = torch.full((M,), 1.0, device=dev) if colsum else None
	ops.gemm(A, B, c, M, N, K, M if a_mn else K, N if b_mn else K, N, a_layout=L.MN_MAJOR if a_mn else L.K_MAJOR, b_layout=L.MN_MAJOR if b_mn else L.K_MAJOR,
		accumulate=acc, a_colsum=cs, workspace=ws)
	kern = lib.vitk_gemm_last_kernel().decode()
	seen[kern] = seen.get(kern, 0) + 1
	want = (ref + (2.0 if acc else 0.0)).to(c.dtype).double()
	if not torch.equal(c.cpu().double(), want) or (colsum and not torch.equal(cs.cpu().double(), 1.0 + a.double().sum(1))):
		print("MISMATCH", dict(case=case, M=M, N=N, K=K, a_mn=a_mn, b_mn=b_mn, out_f32=out_f32, acc=acc, colsum=colsum, ws=ws, kernel=kern), flush=True)
		sys.exit(1)
	# a group of weight-gradient-shaped products that share the scratch buffer
	if case % 5 == 0:
		n = rng.randint(2, 10)
		held, wants, outs = [], [], []
		for i in range(n):
			Mi, Ni = rng.choice([64, 72, 192, 264, 384, 576, 768]), rng.choice([64, 192, 200, 384, 768])
			Ki = rng.choice([1024, 2048, 4096, 16384, 33280, 65536])
			gi = torch.Generator().manual_seed(1000 * case + i)
			ai, Ai = operand(Mi, Ki, True, gi)
			bi, Bi = operand(Ni, Ki, True, gi)
			acc_i, cs_i = rng.random() < 0.5, rng.random() < 0.5
			ci = torch.full((Mi, Ni), 3.0 if acc_i else float("nan"), device=dev)
			csum = torch.full((Mi,), 1.0, device=dev) if cs_i else None
			ops.gemm(Ai, Bi, ci, Mi, Ni, Ki, Mi, Ni, Ni, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR, accumulate=acc_i, a_colsum=csum, hold=held)
			outs.append((ci, csum))
			wants.append((ai.double() @ bi.double().t() + (3.0 if acc_i else 0.0), 1.0 + ai.double().sum(1)))
		descs = (L.GemmDesc * n)(*[h[0] for h in held])
		L.check(lib.vitk_gemm_group(descs, n, L.stream()))
		kern = lib.vitk_gemm_last_kernel().decode()
		seen["group:" + kern] = seen.get("group:" + kern, 0) + 1
		for i, ((ci, csum), (wc, wcs)) in enumerate(zip(outs, wants)):
			if not torch.equal(ci.cpu().double(), wc) or (csum is not None and not torch.equal(csum.cpu().double(), wcs)):
				print("GROUP MISMATCH", dict(case=case, i=i, n=n, shapes=[(h[0].M, h[0].N, h[0].K) for h in held]), flush=True)
				sys.exit(1)
	if case % 20 == 19:
		print(case + 1, "cases ok", seen, flush=True)
print("all ok", seen)
