"""CPU, world_size 2, gloo: the flat-buffer bucketed gradient all-reduce of vit_amd.dp gives every rank the SUM of
the per-rank gradients, i.e. (after the optimizer's 1/world scale) the gradient of the global batch, for any bucket
size, including unused parameters.  The collective path is the same code that runs over RCCL on the GPUs."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, bucket_bytes, ret, bf16=False):
	sys.path.insert(0, ROOT)
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	from vit_amd import dp
	torch.manual_seed(0)
	model = nn.Sequential(nn.Linear(16, 32), nn.GELU(), nn.Linear(32, 32), nn.GELU(), nn.Linear(32, 4))
	unused = nn.Linear(3, 3)  # never used in forward: its bucket must still be reduced (zeros)
	model.add_module("unused", unused)
	flat = dp.FlatModel(model, bucket_bytes=bucket_bytes)
	sync = dp.GradSync(flat, bf16_buckets=bf16)
	g = torch.Generator().manual_seed(42)
	x_all = torch.randn(8, 16, generator=g)
	y_all = torch.randn(8, 4, generator=g)
	xs, ys = x_all[rank::world], y_all[rank::world]
	results = []
	for step in range(2):
		flat.zero_grad()
		out = model[4](model[3](model[2](model[1](model[0](xs)))))
		loss = ((out - ys) ** 2).sum()
		loss.backward()
		sync.finish()
		results.append(flat.grad.clone())
	# single-process reference on the global batch
	ref = nn.Sequential(nn.Linear(16, 32), nn.GELU(), nn.Linear(32, 32), nn.GELU(), nn.Linear(32, 4))
	torch.manual_seed(0)
	ref = nn.Sequential(nn.Linear(16, 32), nn.GELU(), nn.Linear(32, 32), nn.GELU(), nn.Linear(32, 4))
	out = ref(x_all)
	((out - y_all) ** 2).sum().backward()
	ok = True
	for name, p in ref.named_parameters():
		i = flat.names.index(name)
		o = flat.offsets[i]
		got = results[-1][o:o + p.numel()].view(p.shape)
		if bf16:  # each rank's share is rounded to bfloat16 (2^-9 relative) and the sum once more
			ok = ok and float((got - p.grad).abs().max()) <= 1.5e-2 * float(p.grad.abs().max()) + 1e-6
		else:
			ok = ok and torch.allclose(got, p.grad, rtol=1e-5, atol=1e-6)
	n_buckets = len(flat.buckets)
	all_views = all(p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
	ret[rank] = (ok, n_buckets, all_views, torch.equal(results[0], results[1]))
	dist.destroy_process_group()


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("bucket_bytes", [1 << 30, 2048, 64])
def test_bucketed_allreduce_two_ranks(bucket_bytes, bf16):
	"""bf16: the buckets travel as bfloat16 (GradSync(bf16_buckets=True)); the float32 gradient that comes back is the
	global-batch gradient to bfloat16 rounding."""
	world = 2
	port = 29500 + (os.getpid() % 1000) + {1 << 30: 0, 2048: 1, 64: 2}[bucket_bytes] + (3 if bf16 else 0)
	mgr = mp.Manager()
	ret = mgr.dict()
	mp.spawn(_worker, args=(world, port, bucket_bytes, ret, bf16), nprocs=world, join=True)
	assert len(ret) == world
	for rank in range(world):
		ok, n_buckets, all_views, same = ret[rank]
		assert ok, f"rank {rank}: reduced gradients differ from the global-batch gradients"
		assert all_views, "parameter .grad tensors must stay views of the flat buffer"
		assert same, "two identical steps must give identical reduced gradients"
	if bucket_bytes == 64:
		assert ret[0][1] >= 4
	if bucket_bytes == 1 << 30:
		assert ret[0][1] == 1


def _bcast_worker(rank, world, port, ret):
	sys.path.insert(0, ROOT)
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	from vit_amd import dp
	torch.manual_seed(100 + rank)  # replicas start DIFFERENT
	model = nn.Sequential(nn.Linear(6, 8), nn.BatchNorm1d(8), nn.Linear(8, 3))
	model[2].bias.requires_grad_(False)  # frozen: not in the flat buffer
	with torch.no_grad():
		model[1].running_mean.add_(float(rank) + 0.5)
	flat = dp.FlatModel(model)
	dp.broadcast_replica(flat)
	inside = all(flat.param.data_ptr() <= p.data_ptr() < flat.param.data_ptr() + 4 * flat.numel for p in flat.params)
	ret[rank] = ({k: v.clone() for k, v in model.state_dict().items()}, inside)
	dist.destroy_process_group()


def test_replicas_start_from_rank0():
	world = 2
	port = 29500 + (os.getpid() % 1000) + 7
	mgr = mp.Manager()
	ret = mgr.dict()
	mp.spawn(_bcast_worker, args=(world, port, ret), nprocs=world, join=True)
	torch.manual_seed(100)
	want = nn.Sequential(nn.Linear(6, 8), nn.BatchNorm1d(8), nn.Linear(8, 3))
	with torch.no_grad():
		want[1].running_mean.add_(0.5)
	for rank in range(world):
		sd, inside = ret[rank]
		assert inside, "parameters must stay views of the flat buffer after the broadcast"
		for k, v in want.state_dict().items():
			assert torch.equal(sd[k], v), f"rank {rank}: {k} is not rank 0's"


def test_flat_model_layout():
	sys.path.insert(0, ROOT)
	from vit_amd import dp
	m = nn.Sequential(nn.Linear(5, 7), nn.LayerNorm(7), nn.Linear(7, 3))
	before = {k: v.clone() for k, v in m.state_dict().items()}
	flat = dp.FlatModel(m, bucket_bytes=64)
	for k, v in m.state_dict().items():
		assert torch.equal(v, before[k])
	assert all(o % 64 == 0 for o in flat.offsets)
	covered = sorted((s, e) for s, e, _ in flat.buckets)
	assert covered[0][0] == 0 and covered[-1][1] == flat.numel
	assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
	assert dp.no_decay("blocks.0.attn_norm.weight") and dp.no_decay("mlp.bias") and not dp.no_decay("blocks.0.attn_layer_scale")


def _shared_worker(rank, world, port, ret):
	sys.path.insert(0, ROOT)
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	from vit_amd import dp

	def build():
		torch.manual_seed(0)
		return nn.ModuleDict(dict(inp=nn.Linear(16, 32), shared=nn.Linear(32, 32), a=nn.Linear(32, 4), b=nn.Linear(32, 4)))

	def fwd(m, x, branch):
		h = torch.tanh(m["inp"](x))
		h = torch.tanh(m["shared"](torch.tanh(m["shared"](h))))  # one module, two uses: its gradient is written twice
		return m[branch](h)  # the other head stays unused in this rank's graph

	model = build()
	flat = dp.FlatModel(model, bucket_bytes=64)  # many buckets
	sync = dp.GradSync(flat)
	g = torch.Generator().manual_seed(7)
	x_all, y_all = torch.randn(8, 16, generator=g), torch.randn(8, 4, generator=g)
	xs, ys = x_all[rank::world], y_all[rank::world]
	branch = "a" if rank == 0 else "b"  # the ranks' sets of unused parameters differ
	grads, early = [], []
	for step in range(3):
		flat.zero_grad()
		if step < 2:
			((fwd(model, xs, branch) - ys) ** 2).sum().backward()
		else:  # gradient accumulation: two micro-batches, only the second synchronises
			with sync.no_sync():
				((fwd(model, xs[:2], branch) - ys[:2]) ** 2).sum().backward()
			((fwd(model, xs[2:], branch) - ys[2:]) ** 2).sum().backward()
		early.append(len(sync.launched))
		sync.finish()
		grads.append(flat.grad.clone())
	ref = build()
	for r in range(world):
		((fwd(ref, x_all[r::world], "a" if r == 0 else "b") - y_all[r::world]) ** 2).sum().backward()
	ok = []
	for name, p in ref.named_parameters():
		i = flat.names.index(name)
		o = flat.offsets[i]
		want = p.grad if p.grad is not None else torch.zeros_like(p)
		ok.append(all(torch.allclose(gr[o:o + p.numel()].view(p.shape), want, rtol=1e-5, atol=1e-6) for gr in grads))
	ret[rank] = (all(ok), early, len(flat.buckets), sync.expected[flat.names.index("shared.weight")])
	dist.destroy_process_group()


def test_shared_module_microbatches_and_rank_dependent_unused_parameters():
	world = 2
	port = 29500 + (os.getpid() % 1000) + 11
	mgr = mp.Manager()
	ret = mgr.dict()
	mp.spawn(_shared_worker, args=(world, port, ret), nprocs=world, join=True)
	for rank in range(world):
		ok, early, n_buckets, shared_expected = ret[rank]
		assert ok, f"rank {rank}: reduced gradients differ from the global-batch gradients"
		assert shared_expected == 1 or shared_expected == 2  # autograd accumulates both uses before its hook fires once
		assert early[0] == 0, "the first backward only learns the write counts"
		assert n_buckets >= 4


def _single_rank_worker(rank, world, port, ret):
	sys.path.insert(0, ROOT)
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	from vit_amd import dp
	torch.manual_seed(0)
	model = nn.Sequential(nn.Linear(16, 32), nn.GELU(), nn.Linear(32, 4))
	flat = dp.FlatModel(model, bucket_bytes=64)
	idle = dp.GradSync(flat)  # one rank, default: nothing to do, no hooks
	assert not idle.active and all(h is None for h in idle.hooks)
	sync = dp.GradSync(flat, single_rank=True)
	x = torch.randn(8, 16, generator=torch.Generator().manual_seed(1))
	launched = []
	for step in range(3):
		flat.zero_grad()
		model(x).square().sum().backward()
		early = list(sync.launched)  # buckets that went out during backward (none in the first step: the counts are learnt there)
		sync.finish()
		launched.append(early)
	ref = nn.Sequential(nn.Linear(16, 32), nn.GELU(), nn.Linear(32, 4))
	ref.load_state_dict(model.state_dict())
	ref(x).square().sum().backward()
	same = all(torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-7) for p, q in zip(model.parameters(), ref.parameters()))
	ret[rank] = (same, len(flat.buckets), launched)
	dist.destroy_process_group()


def test_single_rank_group_runs_the_collective_sequence():
	"""GradSync(single_rank=True) in a process group of ONE rank (what tools/rccl_single_rank.py runs over RCCL on one GPU):
	hooks are installed, every bucket is all-reduced (the identity), in bucket order and from the second step on during
	backward, and the gradients are those of a plain backward pass."""
	port = 29500 + (os.getpid() % 1000) + 7
	mgr = mp.Manager()
	ret = mgr.dict()
	mp.spawn(_single_rank_worker, args=(1, port, ret), nprocs=1, join=True)
	same, n_buckets, launched = ret[0]
	assert same and n_buckets >= 3
	assert launched[0] == []  # first step: everything goes out in finish()
	assert launched[1] == list(range(n_buckets)) and launched[2] == launched[1]
