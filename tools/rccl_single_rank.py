"""The multi-GPU step's RCCL call sequence on ONE GPU: a process group of one rank (backend "nccl" = RCCL), gradient
buckets all-reduced in place from the backward hooks, waits, fused AdamW.  Every collective is the identity there, so
the parameters after a few steps must equal those of the same steps without any collective; the script also times the
ViT-B/16 step both ways.  Start it under the launcher (it spawns the rank before anything touches the GPU):
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_single_rank.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_world_size() == 1
from vit_amd import dp, ops  # noqa: E402
from vit_amd.models import ViT  # noqa: E402


def run(cfg, batch, steps, sync, timed=False):
	ops.clear_grad_sinks()
	torch.manual_seed(0)
	model = ViT(bench.make_config(cfg)).to(dev)
	trainer = dp.Trainer(model, lr=1e-3, weight_decay=0.05, single_rank_sync=sync)
	g = torch.Generator().manual_seed(7)
	x = torch.randn(batch, *cfg["chw"], generator=g).to(dev)
	y = torch.randint(0, cfg["out_d"], (batch,), generator=g).to(dev)
	losses = []

	def step():
		trainer.zero_grad()
		with torch.autocast("cuda", dtype=torch.bfloat16):
			loss = ops.cross_entropy(model(x), y)
		loss.backward()
		trainer.step()
		return loss
	for _ in range(steps):
		losses.append(float(step()))
	ms = None
	if timed:
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(10):
			step()
		torch.cuda.synchronize()
		ms = (time.perf_counter() - t0) / 10 * 1e3
	out = (losses, trainer.flat.param.clone(), len(trainer.flat.buckets), list(trainer.sync.launched), ms)
	trainer.close()
	return out


ok = True
for name, batch in (("vit_tiny", 64), ("vit_b16", 32)):
	cfg = bench.CONFIGS[name]
	l0, p0, nb, _, _ = run(cfg, batch, 4, False)
	l1, p1, nb1, launched, _ = run(cfg, batch, 4, True)
	err = ((p1 - p0).abs().max() / p0.abs().max()).item()
	print(f"{name}: {nb1} buckets, losses without / with the collectives {l0} / {l1}, max parameter difference {err:.2e}", flush=True)
	ok = ok and err < 1e-3 and all(abs(a - b) < 1e-2 * abs(a) for a, b in zip(l0, l1))
cfg = bench.CONFIGS["vit_b16"]
_, _, _, _, ms0 = run(cfg, cfg["batch"], 3, False, timed=True)
_, _, nb, _, ms1 = run(cfg, cfg["batch"], 3, True, timed=True)
print(f"vit_b16 batch {cfg['batch']}: {ms0:.2f} ms/step without, {ms1:.2f} ms/step with {nb} in-place bucket all-reduces per step (one rank)", flush=True)
dist.destroy_process_group()
print("OK" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
