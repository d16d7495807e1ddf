import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch, torch.distributed as dist
from test_gpu_dp import _model, _Done
from vit_amd import dp, ops
gpu = torch.device("cuda:0")
ops.WGRAD_OVERLAP = True; ops.WGRAD_OVERLAP_MIN_FLOPS = 0.0
records = []
def fake_all_reduce(t, op=None, group=None, async_op=False):
	cur = torch.cuda.current_stream()
	cur.synchronize()
	lane = ops._wgrad_lanes.get(0)
	records.append((t.data_ptr(), t.numel(), t.detach().clone(), lane is not None and cur == lane.stream, trig[0]))
	t.mul_(2.0)
	return _Done()
dist.is_initialized = lambda: True
dist.get_world_size = lambda group=None: 2
dist.broadcast = lambda *a, **k: None
dist.all_reduce = fake_all_reduce
model = _model(gpu)
trainer = dp.Trainer(model, lr=1e-2, weight_decay=0.05, bucket_bytes=64 << 10)
trig = [None]
for i, h in enumerate(trainer.sync.hooks):
	def mk(i, h):
		def hook(p):
			trig[0] = trainer.flat.names[i]
			return h(p)
		return hook
	hk = mk(i, h)
	trainer.sync.hooks[i] = hk
	p = trainer.flat.params[i]
	ops.register_grad_sink(p, trainer.flat.grad[trainer.flat.offsets[i]:trainer.flat.offsets[i] + p.numel()].view(p.shape), hk)
g = torch.Generator().manual_seed(7)
x = torch.randn(8, 3, 16, 16, generator=g).to(gpu); y = torch.randint(0, 10, (8,), generator=g).to(gpu)
base = trainer.flat.grad.data_ptr()
print("buckets", [(s, e, [trainer.flat.names[i] for i in idx]) for s, e, idx in trainer.flat.buckets])
for step in range(3):
	records.clear()
	trainer.zero_grad()
	with torch.autocast("cuda", dtype=torch.bfloat16):
		loss = ops.cross_entropy(model(x), y)
	loss.backward()
	nb = len(records)
	trainer.sync.finish()
	torch.cuda.synchronize()
	for k, (p, n, snap, on_lane, who) in enumerate(records):
		s = (p - base) // 4
		ok = torch.equal(trainer.flat.grad[s:s + n], 2.0 * snap)
		print(step, "bucket at", s, "during backward" if k < nb else "in finish", "on lane" if on_lane else "on main", "triggered by", who, "OK" if ok else "EARLY", float(snap.abs().max()))
	trainer.opt.step(grad_scale=0.5)
