"""Headline benchmark: train images/sec of ViT-B/16 at 224x224, bf16 compute, on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config vit_b16|vit_tiny]

For N > 1 it runs one rank per GPU over RCCL: either started under a launcher that sets WORLD_SIZE / RANK / LOCAL_RANK
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`), or started plainly, in which case it
launches the N ranks itself as child processes (before anything in this process touches a GPU) and relays rank 0's
line and the exit code.  A step = forward + cross-entropy + backward + gradient all-reduce (N > 1) + fused AdamW on
synthetic data already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
	# BASELINE config 4: 86,538,472 parameters, 105.38 GFLOP per image per training step (3 x forward, 2 x MAC)
	"vit_b16": dict(d_model=768, n_head=12, n_block=12, n_patch=14, d_ff=3072, out_d=1000, chw=(3, 224, 224), batch=256),
	# BASELINE config 1 geometry (ViT-Tiny patch 4 on 32x32)
	"vit_tiny": dict(d_model=192, n_head=3, n_block=12, n_patch=8, d_ff=768, out_d=10, chw=(3, 32, 32), batch=512),
}
# BASELINE config 3: the reference's CvTConfig defaults (config.py:53-98: CvT-13 geometry 64/192/384, depths 1/2/10, heads 1/3/6,
# SiLU, RMSNorm) on 32x32 inputs with 10 classes
CVT = dict(chw=(3, 32, 32), out_d=10, batch=256, use_layer_scale=False, activation="silu",
	s1=(7, 4, 2, 64, 1, 1, 4.0, 3, 1, 1), s2=(3, 2, 1, 192, 2, 3, 4.0, 3, 2, 1), s3=(3, 2, 1, 384, 10, 6, 4.0, 3, 2, 1))
# BASELINE config 2: DeiT-style distillation of a ViT-Tiny student (with distillation token and head) from a frozen
# CIFAR ResNet-20 teacher (deit_trainer.py:23-52, main.py:106-109 there), alpha 0.5, tau 1, batch 512
DEIT = dict(CONFIGS["vit_tiny"], batch=512, alpha=0.5, tau=1.0)
# BASELINE config 5: the conv + attention hybrid of the reference's older snapshot (_single_file.py:399-458, configuration
# 562-570: dims 64/96/192/384/768, depths 2/2/3/5, 32 heads, SiLU, RMSNorm) at 224x224, 1000 classes, batch 128 per GPU
COATNET = dict(dims=[64, 96, 192, 384, 768], depths=[2, 2, 3, 5, 2], num_heads=32, activation="silu", chw=(3, 224, 224), out_d=1000, batch=128)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak of one MI355X


def train_flops_per_image(c, class_rows_last_block=False):
	"""3 x the forward's 2 x MAC count.  class_rows_last_block: the last block computes only what the class-token row of its
	output depends on (keys / values of all tokens; query, attention row, projection and feed-forward of one row)."""
	d, f, L_ = c["d_model"], c["d_ff"], c["n_block"]
	n = c["n_patch"] ** 2 + 1
	p = c["chw"][1] // c["n_patch"]
	block = 2 * n * d * 3 * d + 4 * n * n * d + 2 * n * d * d + 4 * n * d * f
	last = 2 * n * d * 2 * d + 2 * d * d + 4 * n * d + 2 * d * d + 4 * d * f if class_rows_last_block else block
	fwd = (L_ - 1) * block + last + 2 * (n - 1) * (c["chw"][0] * p * p) * d + 2 * d * c["out_d"]
	return 3.0 * fwd


def make_config(c, dropout=0.0):
	from vit_amd.layers import MHA

	class Cfg:
		attention = MHA
		norm = nn.LayerNorm
		activation = "gelu"
		use_layer_scale = False
		use_qk_norm = False
		dropout = 0.0
	for k in ("d_model", "n_head", "n_block", "n_patch", "d_ff", "out_d", "chw"):
		setattr(Cfg, k, c[k])
	Cfg.dropout = dropout
	return Cfg()


def make_cvt_config(dropout=0.0):
	from vit_amd.layers import RMSNorm

	class Cfg:
		norm = RMSNorm
	Cfg.chw, Cfg.out_d, Cfg.use_layer_scale, Cfg.activation, Cfg.dropout = CVT["chw"], CVT["out_d"], CVT["use_layer_scale"], CVT["activation"], dropout
	names = ("emb_kernel", "emb_stride", "emb_pad", "emb_dim", "depth", "heads", "mlp_ratio", "qkv_kernel", "qkv_stride", "qkv_pad")
	for s_ in (1, 2, 3):
		for n, v in zip(names, CVT[f"s{s_}"]):
			setattr(Cfg, f"s{s_}_{n}", v)
	return Cfg()


def cpu_baseline(c, seconds=20.0):
	"""The CPU oracle (explicit-formula restatement of the reference modules, float32, torch autograd for the
	backward) timed on this host: forward + cross entropy + backward at a small batch, bounded to ~`seconds`."""
	from oracle import vit_oracle as O
	from vit_amd.models import ViT
	torch.manual_seed(0)
	model = ViT(make_config(c))
	sd = {k: v.detach() for k, v in model.state_dict().items()}
	batch = 8 if c["d_model"] >= 768 else 128
	x = torch.randn(batch, *c["chw"])
	y = torch.randint(0, c["out_d"], (batch,))
	fwd = lambda im, s: O.vit_fwd(im, s, c["n_patch"], c["n_block"], c["n_head"], "layernorm", "gelu")
	cores = torch.get_num_threads()
	times = []
	t_begin = time.time()
	O.loss_and_grads(fwd, sd, x, y)  # warm-up
	while len(times) < 5 and (time.time() - t_begin) < seconds:
		t0 = time.time()
		O.loss_and_grads(fwd, sd, x, y)
		times.append(time.time() - t0)
	med = sorted(times)[len(times) // 2]
	return {"value": round(batch / med, 2), "unit": "images/sec", "cores": cores, "kind": "port",
		"sample": f"{len(times)} timed fwd+loss+bwd iterations (no AdamW step) of the float32 CPU oracle at batch {batch} (same model config), "
			"median; the GPU figure is fwd+loss+bwd+AdamW at the per-GPU batch of `config`"}


def launch_ranks(n, argv):
	"""Start `n` ranks of this script under torch.distributed.run as CHILD processes (this process has not touched a GPU
	and never will), pass their stderr through, relay rank 0's JSON line and return the launcher's exit code."""
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		port = s.getsockname()[1]
	env = dict(os.environ)
	env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
	env.setdefault("OMP_NUM_THREADS", "8")
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
		"--master-port", str(port), os.path.abspath(__file__)] + list(argv)
	proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
	lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
	for ln in proc.stdout.splitlines():
		if not ln.startswith("{"):
			print(ln, file=sys.stderr)
	if lines:
		print(lines[-1], flush=True)
	if proc.returncode == 0 and not lines:
		print("bench.py: the ranks exited without a result line", file=sys.stderr)
		return 1
	return proc.returncode


def launcher_selftest(world, rank):
	"""What --launcher-selftest runs in each rank: no GPU, a gloo all-reduce, rank 0 prints a line (CPU test of launch_ranks)."""
	dist.init_process_group("gloo")
	t = torch.tensor([float(rank + 1)])
	dist.all_reduce(t)
	if rank == 0:
		print(json.dumps({"selftest": True, "n_gpus": world, "sum": t.item()}), flush=True)
	dist.destroy_process_group()


def config_dict(name):
	return CVT if name == "cvt" else (DEIT if name == "deit_tiny" else (COATNET if name == "coatnet" else CONFIGS[name]))


def build_workload(name, dropout, dev, lr=1e-4, pipelined=True, bf16_buckets=False):
	"""Model + trainer + loss closure of one named configuration on `dev`: (model, trainer, loss_of(x, y), teacher source)."""
	from vit_amd import dp, ops
	from vit_amd.models import ViT
	c = config_dict(name)
	torch.manual_seed(0)
	teacher, tsrc = None, None
	if name == "cvt":
		from vit_amd.models import CvT
		model = CvT(make_cvt_config(dropout)).to(dev)
	elif name == "coatnet":
		from vit_amd.layers import RMSNorm
		from vit_amd.models import CoAtNet

		class CoatCfg:
			norm = RMSNorm
		CoatCfg.dropout = dropout
		for k_ in ("dims", "depths", "num_heads", "activation", "chw", "out_d"):
			setattr(CoatCfg, k_, c[k_])
		model = CoAtNet(CoatCfg()).to(dev)
	elif name == "deit_tiny":
		from vit_amd.models import DeiTViT
		from vit_amd.models.teacher import frozen_teacher
		model = DeiTViT(make_config(c, dropout)).to(dev)
		# the teacher's weights: the reference's resnet20_cifar10 state_dict as recorded in the golden fixture (trained fc kept)
		tsd, tsrc = None, "random init"
		gpath = os.path.join(ROOT, "tests", "golden", "resnet20_teacher.npz")
		if os.path.exists(gpath):
			import numpy as np
			z = np.load(gpath, allow_pickle=False)
			tsd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
			tsrc = "reference resnet20_cifar10 weights (tests/golden/resnet20_teacher.npz)"
		teacher = frozen_teacher(20, c["out_d"], tsd, fold_bn=True).to(dev)
	else:
		model = ViT(make_config(c, dropout)).to(dev)
	trainer = dp.Trainer(model, lr=lr, weight_decay=0.05, pipelined_step=pipelined, bf16_buckets=bf16_buckets)
	if teacher is not None:
		from vit_amd.models.teacher import TeacherBeside
		beside = TeacherBeside(teacher)  # (the teacher's forward runs on a stream of its own beside the student's)

	def loss_of(xi, yi):
		if teacher is not None:
			# teacher forward under no_grad on stock PyTorch-ROCm in float32, as the reference runs it (deit_trainer.py:33 there;
			# with the BatchNorms folded: 1.3 ms at batch 512 and the reference's logits to 6e-7, where bf16 autocast with the
			# BatchNorm modules took 4.6 ms and was 4.7e-2 off), then the fused two-head loss
			beside.start(xi)
		with torch.autocast("cuda", dtype=torch.bfloat16):
			if teacher is not None:
				cls, dist_ = model(xi)
				t_logits = beside.logits()
				return ops.deit_dual_loss(cls, dist_, yi, t_logits, alpha=c["alpha"], tau=c["tau"])
			return ops.cross_entropy(model(xi), yi)
	return model, trainer, loss_of, tsrc


def time_secondary(name, dev, batch=0, dropout=0.0, steps=10, warmup=3, graph=False):
	"""One further configuration timed in this process behind the headline: {config, per_gpu_batch, ms_per_step, value}."""
	from vit_amd import dp, ops
	c = config_dict(name)
	batch = batch or c["batch"]
	ops.clear_grad_sinks()
	model, trainer, loss_of, _ = build_workload(name, dropout, dev, pipelined=not graph)
	g = torch.Generator(device="cpu").manual_seed(1234)
	x = torch.randn(batch, *c["chw"], generator=g).to(dev)
	y = torch.randint(0, c["out_d"], (batch,), generator=g).to(dev)
	graphed = None
	if graph:
		graphed = dp.GraphedStep(trainer, loss_of, [x, y], warmup=warmup)

	def step():
		if graphed is not None:
			return graphed(x, y)
		trainer.zero_grad()
		loss = loss_of(x, y)
		loss.backward()
		trainer.step()
		return loss
	for _ in range(warmup):
		step()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(steps):
		loss = step()
	torch.cuda.synchronize()
	el = time.perf_counter() - t0
	final = loss.item()
	trainer.wait_step()
	if graphed is not None:
		graphed.close()
	trainer.close()
	ops.clear_grad_sinks()
	del model, trainer, x, y
	torch.cuda.empty_cache()
	return {"config": name, "per_gpu_batch": batch, "dropout": dropout, "hip_graph": graph, "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * el / steps, 3),
		"value": round(batch * steps / el, 1), "unit": "images/sec", "final_loss": round(final, 4)}


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=10)
	ap.add_argument("--warmup", type=int, default=3)
	ap.add_argument("--graph", action="store_true", help="capture the training step into a HIP graph and replay it (single GPU)")
	ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
	ap.add_argument("--config", default="vit_b16", choices=sorted(CONFIGS) + ["cvt", "deit_tiny", "coatnet"])
	ap.add_argument("--dropout", type=float, default=0.0, help="config.dropout (attention-probability, branch and feed-forward dropouts)")
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--no-kernel-timing", action="store_true")
	ap.add_argument("--full-last-block", action="store_true", help="ViT: run the last block on all token rows (default: only what the class-token row, the one row the head reads, depends on; same logits and gradients)")
	ap.add_argument("--no-pipelined-step", action="store_true", help="finish the optimizer step before the next forward starts (default: it runs under it, on the second stream)")
	ap.add_argument("--lr", type=float, default=1e-4)
	ap.add_argument("--bf16-buckets", action="store_true", help="N > 1: all-reduce the gradient buckets as bfloat16 (half the bytes on the links)")
	ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configurations and the per-GPU batch sweep that a default one-GPU run times behind the headline")
	ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
	args = ap.parse_args()

	if "WORLD_SIZE" not in os.environ and args.gpus > 1:
		raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	if world != args.gpus:
		raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
	if args.launcher_selftest:
		return launcher_selftest(world, rank)
	torch.cuda.set_device(local_rank)
	dev = torch.device("cuda", local_rank)
	rccl_ranks = None
	if world > 1:
		os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
		dist.init_process_group("nccl", device_id=dev)
		# how many ranks the collectives of this run really span: a sum of ones over RCCL, reported in the result line
		ones = torch.ones(1, dtype=torch.float32, device=dev)
		dist.all_reduce(ones)
		rccl_ranks = int(ones.item())
		assert rccl_ranks == world, f"RCCL all-reduce saw {rccl_ranks} ranks, WORLD_SIZE is {world}"

	from vit_amd import dp, ops
	import vit_amd.models.vit as vit_model
	if args.full_last_block:
		vit_model.CLASS_ROWS_LAST_BLOCK = False

	is_cvt = args.config == "cvt"
	is_deit = args.config == "deit_tiny"
	is_coat = args.config == "coatnet"
	c = config_dict(args.config)
	batch = args.batch or c["batch"]
	model, trainer, loss_of, tsrc = build_workload(args.config, args.dropout, dev, lr=args.lr, pipelined=not args.no_pipelined_step and not args.graph, bf16_buckets=args.bf16_buckets)
	g = torch.Generator(device="cpu").manual_seed(1234 + rank)
	x = torch.randn(batch, *c["chw"], generator=g).to(dev)
	y = torch.randint(0, c["out_d"], (batch,), generator=g).to(dev)

	def step():
		trainer.zero_grad()
		loss = loss_of(x, y)
		loss.backward()
		trainer.step()
		return loss

	def barrier():
		if world > 1:
			dist.barrier()
		torch.cuda.synchronize()

	graphed = None
	if args.graph:
		# the whole step as one HIP graph launch: for launch-bound configurations (ViT-Tiny, CvT).  With more than one rank the
		# graph holds forward + backward and the gradient all-reduces + optimizer step follow it eagerly (dp.GraphedStep)

		graphed = dp.GraphedStep(trainer, loss_of, [x, y], warmup=max(args.warmup, 3))

		def step():
			return graphed(x, y)
	first = None  # loss of the very first step (from the initial weights): with final_loss it shows the optimizer at work
	for _ in range(args.warmup):
		l0 = step()
		if first is None:
			first = l0.detach().clone()
	barrier()
	t0 = time.perf_counter()
	for _ in range(args.steps):
		loss = step()
	barrier()
	elapsed = time.perf_counter() - t0
	profiled_steps = 0
	if not args.no_kernel_timing and rank == 0 and world == 1 and graphed is None and not (is_cvt or is_deit or is_coat):
		# per-launch HIP events for the roofline object (the ViT configurations print one), in a pass of their own BEHIND the
		# timed steps (the ~2000 event records cost the step 0.6 ms), the same steps on the same data with every GEMM on
		# the main stream: a launch that shares the CUs with a weight-gradient GEMM of the second stream has no duration of
		# its own to divide its FLOPs by
		ops.GEMM_PROFILE, ops.GEMM_PROFILE_SERIAL = [], True
		trainer.pipelined_step = False  # (the optimizer does not share the CUs with the next forward's GEMMs either)
		trainer.wait_step()
		profiled_steps = min(args.steps, 10)
		for _ in range(profiled_steps):
			step()
		torch.cuda.synchronize()
		ops.GEMM_PROFILE_SERIAL = False
	t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
	if world > 1:
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
	elapsed = t.item()
	final_loss = loss.item()
	first_loss = first.item() if first is not None else None
	assert final_loss == final_loss, "loss is NaN"

	if rank == 0 and is_deit:
		print(json.dumps({"metric": "train images/sec (node) DeiT-Tiny distillation 32x32 bf16", "value": round(batch * world * args.steps / elapsed, 2), "unit": "images/sec",
			"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
			"scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
			"config": {"workload": "deit_tiny: ViT-Tiny/4 student with distillation token + head (d192/h3/L12/ff768, 66 tokens) distilled from a frozen CIFAR ResNet-20 "
				f"teacher ({tsrc}; eval BatchNorms folded into the convolutions; float32, stock PyTorch-ROCm, no_grad), alpha {c['alpha']}, tau {c['tau']}, teacher fwd + student fwd + fused two-head loss + bwd + AdamW",
				"global_batch": batch * world, "per_gpu_batch": batch, "parallelism": f"dp{world}", "hip_graph": bool(args.graph), "dropout": args.dropout},
			"first_loss": None if first_loss is None else round(first_loss, 4), "final_loss": round(final_loss, 4), "lr": args.lr, "rccl_ranks": rccl_ranks}), flush=True)
	elif rank == 0 and is_coat:
		print(json.dumps({"metric": "train images/sec (node) CoAtNet hybrid 224x224 bf16", "value": round(batch * world * args.steps / elapsed, 2), "unit": "images/sec",
			"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
			"scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
			"config": {"workload": "coatnet: conv stem + MBConv stages (112^2 x 64, 56^2 x 96, 28^2 x 192) + 3 transformer blocks on 784 tokens x 384 (32 heads of 12) + "
				"5 on 196 tokens x 768 (32 heads of 24), SiLU, RMSNorm, 1000 classes, fwd+CE+bwd+AdamW; throughput only (the reference's class cannot be constructed)",
				"global_batch": batch * world, "per_gpu_batch": batch, "parallelism": f"dp{world}", "hip_graph": bool(args.graph), "dropout": args.dropout},
			"first_loss": None if first_loss is None else round(first_loss, 4), "final_loss": round(final_loss, 4), "lr": args.lr, "rccl_ranks": rccl_ranks}), flush=True)
	elif rank == 0 and is_cvt:
		# secondary configuration (BASELINE config 3): throughput only, no FLOP model / roofline for it
		print(json.dumps({"metric": "train images/sec (node) CvT-13 geometry 32x32 bf16", "value": round(batch * world * args.steps / elapsed, 2), "unit": "images/sec",
			"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
			"scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
			"config": {"workload": "cvt: CvTConfig defaults (64/192/384, depths 1/2/10, heads 1/3/6, silu, RMSNorm) at 32x32, 10 classes, fwd+CE+bwd+AdamW",
				"global_batch": batch * world, "per_gpu_batch": batch, "parallelism": f"dp{world}", "hip_graph": bool(args.graph), "dropout": args.dropout},
			"first_loss": None if first_loss is None else round(first_loss, 4), "final_loss": round(final_loss, 4), "lr": args.lr, "rccl_ranks": rccl_ranks}), flush=True)
	elif rank == 0:
		value = batch * world * args.steps / elapsed
		class_rows = vit_model.CLASS_ROWS_LAST_BLOCK and args.dropout == 0.0
		flops_img = train_flops_per_image(c, class_rows)  # FLOPs the step executes
		flops_nominal = train_flops_per_image(c)  # FLOPs of the reference's formulation (every block on every token row)
		out = {
			"metric": "train images/sec (node) ViT-B/16 224x224 bf16" if args.config == "vit_b16" else "train images/sec (node) ViT-Tiny/4 32x32 bf16",
			"value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
			"ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
			"dtype": "bf16", "data": "synthetic",
			"config": {"workload": f"{args.config}: ViT d{c['d_model']}/h{c['n_head']}/L{c['n_block']}/ff{c['d_ff']} patch{c['chw'][1] // c['n_patch']} at {c['chw'][1]}x{c['chw'][2]}, "
				f"{c['n_patch'] ** 2 + 1} tokens, fwd+CE+bwd+AdamW, fp32 master weights, bf16 activations", "global_batch": batch * world, "per_gpu_batch": batch,
				"parallelism": f"dp{world}", "hip_graph": bool(args.graph), "dropout": args.dropout,
				"gradient_buckets": "bf16" if args.bf16_buckets else "f32",
				"last_block": "class rows only: the head reads out[:, 0], the last block computes what that row depends on (same logits and gradients as the full block; --full-last-block runs all rows)" if class_rows else "all token rows",
				"optimizer_step": "runs under the next forward on the second stream" if not args.no_pipelined_step and not args.graph else "finished before the next forward"},
			"train_gflop_per_image": round(flops_img / 1e9, 3),
			"train_gflop_per_image_all_rows": round(flops_nominal / 1e9, 3),
			"mfma_util_pct": round(100.0 * value * flops_img / (world * PEAK_BF16_TFLOPS * 1e12), 2),
			"first_loss": None if first_loss is None else round(first_loss, 4), "final_loss": round(final_loss, 4), "lr": args.lr,
		}
		if ops.GEMM_PROFILE:
			prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
			agg = {}
			for label, flops, e0, e1 in prof:
				a = agg.setdefault(label, [0.0, 0.0, 0])
				a[0] += flops
				a[1] += e0.elapsed_time(e1) * 1e-3
				a[2] += 1
			dom = max(agg.items(), key=lambda kv: kv[1][1])
			label, (fl, sec, n) = dom
			achieved = fl / sec / 1e12
			# HBM-side bytes per launch of that kernel: PMC counters cannot be read from inside this process; they come from
			# separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (tools/pmc_traffic.py)
			traffic, traffic_src = None, None
			tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r3_pmc_hbm_traffic.json")
			if args.config == "vit_b16" and os.path.exists(tpath):
				rec = json.load(open(tpath)).get(label)
				if rec:
					traffic, traffic_src = round(rec["traffic_bytes_per_step"] * profiled_steps / n), "profiles/r3_pmc_hbm_traffic.json"
			out["roofline"] = {"bound": "mfma", "kernel": label, "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
				"frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "launches": n, "avg_launch_ms": round(1e3 * sec / n, 4),
				"gflop_per_launch": round(fl / n / 1e9, 2),
				"timed_in": f"{profiled_steps} further steps behind the timed ones, HIP events around every GEMM launch, all GEMMs on one stream",
				"traffic_unit": "bytes per vitk_gemm call (HBM-side, PMC)", "traffic_source": traffic_src,
				"all_gemm_kernels": {k: {"tflops": round(v[0] / v[1] / 1e12, 1), "ms_per_step": round(1e3 * v[1] / profiled_steps, 3), "launches_per_step": v[2] // profiled_steps} for k, v in agg.items()}}
			# matrix-pipe busy fraction of the same workload from an SQ counter pass (SQ_VALU_MFMA_BUSY_CYCLES over
			# 4 x SQ_BUSY_CU_CYCLES, i.e. against the clock the chip actually ran at): recorded, not measured in this run
			spath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r3_pmc_sq_counters.json")
			if args.config == "vit_b16" and os.path.exists(spath):
				for key, rec in json.load(open(spath)).items():
					if key == "whole run" and isinstance(rec, dict) and "derived_mfma_pipe_busy_frac" in rec:
						out["roofline"]["mfma_pipe_busy_frac_whole_step_pmc"] = round(rec["derived_mfma_pipe_busy_frac"], 4)
						out["roofline"]["mfma_pipe_busy_source"] = "profiles/r3_pmc_sq_counters.json"
		if rccl_ranks is not None:
			out["rccl_ranks"] = rccl_ranks  # ranks that a sum of ones over the RCCL process group counted
		if world == 1 and args.config == "vit_b16" and not args.no_secondary and graphed is None and not args.batch and args.dropout == 0.0:
			# behind the headline (whose fields above are final): the other BASELINE configurations and the headline model at
			# the reference's default dropout, ~10 steps each, and the headline model over per-GPU batches (only multiples of 256
			# images give whole 256-row tiles: 197 tokens per image is odd)
			trainer.wait_step()
			trainer.close()
			ops.clear_grad_sinks()
			del model, trainer, x, y, loss, loss_of
			torch.cuda.empty_cache()
			secondary, sweep = [], []
			# (the small configurations take 5-7 ms a step and are host-bound when eager: 30 steps behind 5 warm-up steps, as the
			# allocator and the launch path need more than three steps to settle)
			quick = dict(steps=30, warmup=5)
			runs = [("vit_tiny", dict(quick)), ("vit_tiny", dict(quick, graph=True)), ("deit_tiny", dict(quick)), ("cvt", dict(quick)), ("cvt", dict(quick, graph=True)),
				("coatnet", dict(steps=6)),
				("vit_b16", dict(dropout=0.3, steps=6))]
			for name, kw in runs:
				try:
					secondary.append(time_secondary(name, dev, **kw))
				except Exception as e:  # the headline stands on its own
					secondary.append({"config": name, **kw, "error": repr(e)[:200]})
			for b in (64, 128, 384, 512):
				try:
					r = time_secondary("vit_b16", dev, batch=b, steps=6, warmup=2)
					r["mfma_util_pct"] = round(100.0 * r["value"] * flops_img / (PEAK_BF16_TFLOPS * 1e12), 2)
					sweep.append(r)
				except Exception as e:
					sweep.append({"config": "vit_b16", "per_gpu_batch": b, "error": repr(e)[:200]})
			out["secondary"] = secondary
			out["batch_sweep"] = sweep
		if not args.no_cpu_baseline and world == 1:
			out["cpu_baseline"] = cpu_baseline(c)
		print(json.dumps(out), flush=True)
	if world > 1:
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
