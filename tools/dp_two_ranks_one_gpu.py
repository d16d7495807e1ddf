"""Two (or more) data-parallel ranks: the multi-rank control flow of dp.Trainer -- replica broadcast, gradient buckets
all-reduced from the backward hooks while weight-gradient GEMMs run on the second stream, pipelined optimizer step -- with
a real reduction between processes.  Each rank takes its share of a fixed global batch; after every step the ranks must
hold identical parameters, and these must follow a single-process run over the whole batch.  Two models: the plain ViT
(every gradient arrives through a sink-aware backward function) and the same with LayerScale (those vectors arrive
through autograd's own accumulation hooks, on the main stream).

BACKEND=gloo (default): all ranks on GPU 0, collectives staged through host memory by this tool (below) -- what a
one-GPU box can run; the asynchronous overlap of RCCL collectives with the backward is NOT exercised by it.
BACKEND=nccl: one GPU per rank, torch.distributed "nccl" (= RCCL), the collectives exactly as dp.GradSync issues them
(async_op, waited for in finish()).  Start under the launcher:
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/dp_two_ranks_one_gpu.py"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

import faulthandler  # noqa: E402
faulthandler.dump_traceback_later(int(os.environ.get("STACKS_AFTER", "240")), exit=True)  # a hang names its place and ends the process
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("BACKEND", "gloo")
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
torch.cuda.set_device(dev)
if backend == "nccl":
	os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
	dist.init_process_group("nccl", device_id=dev)
else:
	dist.init_process_group("gloo")
from vit_amd import dp, ops  # noqa: E402
from vit_amd.models import ViT  # noqa: E402


# gloo's own path for device tensors (pinned staging buffers filled and drained by asynchronous copies on streams of its
# own, one per collective) stalls intermittently when two processes share one GPU: tools/diag/gloo_two_ranks_plain.py shows
# it with stock torch modules alone.  The collectives here therefore stage through host memory themselves: the calling
# stream is drained, the payload reduced (or broadcast) on the host, and copied back on the calling stream -- the order
# in which the trainer issues them, and the streams it issues them on, are what this run checks.  GLOO_DEVICE_PATH=1
# hands the device tensors to gloo as they are.
if backend == "gloo" and os.environ.get("GLOO_DEVICE_PATH", "0") == "0":
	class _Done:
		def wait(self, *a):
			return True

		def is_completed(self):
			return True

	def _staged(collective):
		def call(t, *args, async_op=False, **kw):
			if not t.is_cuda:
				return collective(t, *args, async_op=async_op, **kw)
			torch.cuda.current_stream().synchronize()
			h = t.cpu()
			collective(h, *args, **kw)
			t.copy_(h)
			return _Done() if async_op else None
		return call

	dist.all_reduce, dist.broadcast = _staged(dist.all_reduce), _staged(dist.broadcast)

ops.WGRAD_OVERLAP_MIN_FLOPS = float(os.environ.get("MIN_FLOPS", "0"))  # every fused-block weight gradient on the second stream
cfg = dict(bench.CONFIGS["vit_tiny"], n_block=4)
steps, per_rank = 4, 16
g = torch.Generator().manual_seed(3)
xs = torch.randn(steps, world * per_rank, *cfg["chw"], generator=g)
ys = torch.randint(0, cfg["out_d"], (steps, world * per_rank), generator=g)


def make_model(layer_scale):
	c = bench.make_config(cfg)
	c.use_layer_scale = layer_scale
	model = ViT(c)
	if layer_scale:  # (at the initial 1e-4 the blocks' gradients would sit at the noise floor)
		with torch.no_grad():
			for n, p in model.named_parameters():
				if n.endswith("layer_scale"):
					p.fill_(0.5)
	return model.to(dev)


def run(sharded, pipelined, layer_scale):
	ops.clear_grad_sinks()
	torch.manual_seed(100 + (rank if sharded else 0))  # different initial weights per rank: the broadcast must equalise them
	model = make_model(layer_scale)
	if sharded:
		trainer = dp.Trainer(model, lr=1e-3, weight_decay=0.05, bucket_bytes=256 << 10, pipelined_step=pipelined)
	else:
		was, dist.is_initialized = dist.is_initialized, (lambda: False)  # a plain single-process trainer
		try:
			if rank != 0:
				torch.manual_seed(100)
				model = make_model(layer_scale)
			trainer = dp.Trainer(model, lr=1e-3, weight_decay=0.05, pipelined_step=pipelined)
		finally:
			dist.is_initialized = was
	losses = []
	for s in range(steps):
		x, y = xs[s], ys[s]
		if sharded:
			x, y = x[rank * per_rank:(rank + 1) * per_rank], y[rank * per_rank:(rank + 1) * per_rank]
		state.update(trainer=trainer, step=(sharded, pipelined, s), stage="forward")
		trainer.zero_grad()
		with torch.autocast("cuda", dtype=torch.bfloat16):
			loss = ops.cross_entropy(model(x.to(dev)), y.to(dev))
		state["stage"] = "backward"
		loss.backward()
		state["stage"] = "step"
		trainer.step()
		losses.append(float(loss.detach()))
	trainer.wait_step()
	torch.cuda.synchronize()
	out = trainer.flat.param.clone(), len(trainer.flat.buckets), list(trainer.sync.launched), losses
	trainer.close()
	return out


state = {}


def watchdog(after):
	"""After `after` seconds without finishing: which collectives are outstanding and which streams still have work."""
	import threading
	import time

	def body():
		time.sleep(after)
		t = state.get("trainer")
		if t is None:
			return
		lane = ops.second_stream(dev)
		print(f"[rank {rank}] step {state.get('step')} stage {state.get('stage')}: works completed {[w.is_completed() for w in t.sync.works]}, launched {t.sync.launched}, "
			f"next bucket {t.sync.next_bucket} of {len(t.flat.buckets)}, main stream idle {torch.cuda.default_stream(dev).query()}, second stream idle {lane.stream.query()}, "
			f"held {len(lane.held)} hooked {lane.hooked}", flush=True)
	threading.Thread(target=body, daemon=True).start()


if os.environ.get("WATCHDOG"):
	watchdog(float(os.environ["WATCHDOG"]))
ok = True
for layer_scale, pipelined in ((False, False), (False, True), (True, False), (True, True)):
	p_dp, nb, _, l_dp = run(True, pipelined, layer_scale)
	if backend == "nccl":
		gathered = [torch.empty_like(p_dp) for _ in range(world)]
		dist.all_gather(gathered, p_dp)
	else:
		gathered = [torch.empty_like(p_dp).cpu() for _ in range(world)]
		dist.all_gather(gathered, p_dp.cpu())
	same = max(float((t - gathered[0]).abs().max()) for t in gathered)
	p_one, _, _, l_one = run(False, pipelined, layer_scale)
	err = float((p_dp - p_one).abs().max() / p_one.abs().max())
	if rank == 0:
		print(f"pipelined_step={pipelined} layer_scale={layer_scale} backend={backend} world={world}: {nb} buckets, replicas differ by {same:.1e} after {steps} steps, sharded vs single-process parameters {err:.2e}, "
			f"second-stream launches so far {ops.wgrad_sections()}, losses (rank 0 shard / whole batch) {[round(v, 4) for v in l_dp]} / {[round(v, 4) for v in l_one]}", flush=True)
	ok = ok and same == 0.0 and err < 2e-3
dist.barrier()
dist.destroy_process_group()
if rank == 0:
	print("OK" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
