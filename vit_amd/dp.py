"""Batch-sharded (data-parallel) training support: flat parameter / gradient buffers, bucketed gradient
all-reduce overlapped with the backward pass, and the fused AdamW step.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank holds a full
replica; the only exchange per step is the sum of gradients.  Gradients live in ONE flat float32
buffer laid out in parameter-registration order, so a bucket is a contiguous slice that is all-reduced
in place (no packing copies).  Buckets are formed from the END of the buffer (the parameters whose
gradients are produced first in the backward) and are launched from post-accumulate-grad hooks as
soon as their last gradient lands, so the collectives run under the remaining backward kernels.
xGMI is point-to-point: few large buckets (default 48 MiB) keep each ring step bandwidth-bound
rather than latency-bound.

The reference has no distributed code; the single-device semantics it defines (one optimizer step on
the mean gradient of the global batch) are what the sharded step reproduces.
"""
import torch
import torch.distributed as dist

from . import _lib as L
from . import ops


def no_decay(name):
	"""Parameter-group rule of the reference (utils.py:102-121 there): names ending in 'bias' or containing 'norm' or
	'layerscale' (case-insensitive) get no weight decay.  (The block's LayerScale parameters are called
	'*_layer_scale', which that substring test does not match, so they ARE decayed — reproduced as is.)"""
	low = name.lower()
	return name.endswith("bias") or ("norm" in low) or ("layerscale" in low)


def _joint_order(named):
	"""Order of the parameters in the flat buffers: the module's own, except that the q / k / v projection weights of an
	attention module that keeps three Linear layers on one input (layers.attention.ConvAttention; layers/attention.py:57-64
	of the reference) follow one another, then their three biases, then its two depthwise convolution weights.  Each group is
	then ONE matrix in the buffer (sizes that are multiples of 64 elements leave no padding between them), which is what lets
	ops.ConvAttentionFn run the three projections as one GEMM on views of the parameters, of their bf16 copies and of their
	gradient sinks."""
	by_name = dict(named)
	out, used = [], set()
	for n, p in named:
		if n in used:
			continue
		if n.endswith("q_proj.weight"):
			base = n[:-len("q_proj.weight")]
			group = [base + s for s in ("q_proj.weight", "k_proj.weight", "v_proj.weight", "q_proj.bias", "k_proj.bias", "v_proj.bias",
				"conv_proj_k.weight", "conv_proj_v.weight")]
			group = [g for g in group if g in by_name and g not in used]
			out += [(g, by_name[g]) for g in group]
			used.update(group)
			continue
		out.append((n, p))
		used.add(n)
	assert len(out) == len(named)
	return out


class FlatModel:
	"""Moves a module's parameters into one flat float32 buffer (and their .grad into another)."""

	def __init__(self, module, bucket_bytes=48 << 20):
		self.module = module
		in_module_order = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
		named = _joint_order(in_module_order)
		assert named, "no trainable parameters"
		dev = named[0][1].device
		assert all(p.dtype == torch.float32 and p.device == dev for _, p in named), "flat buffers hold float32 parameters on one device"
		# 64-element alignment keeps every parameter 256-byte aligned (16-byte vector loads in the kernels)
		offs, total = [], 0
		for _, p in named:
			offs.append(total)
			total += (p.numel() + 63) // 64 * 64
		self.numel = total
		self.param = torch.zeros(total, dtype=torch.float32, device=dev)
		self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
		self.names = [n for n, _ in named]
		self.params = [p for _, p in named]
		self.offsets = offs
		# indices into names / params / offsets in the module's own named_parameters() order (the order optimizers and
		# checkpoints number parameters in)
		where = {n: i for i, n in enumerate(self.names)}
		self.module_order = [where[n] for n, _ in in_module_order]
		for (n, p), o in zip(named, offs):
			self.param[o:o + p.numel()].copy_(p.data.reshape(-1))
			p.data = self.param[o:o + p.numel()].view(p.shape)
			p.grad = self.grad[o:o + p.numel()].view(p.shape)
		# buckets: contiguous [start, end) ranges, formed from the end
		self.buckets = []  # (start, end, [param indices])
		end, idxs = total, []
		for i in range(len(named) - 1, -1, -1):
			idxs.append(i)
			if (end - offs[i]) * 4 >= bucket_bytes or i == 0:
				self.buckets.append((offs[i], end, idxs))
				end, idxs = offs[i], []
		self.bucket_of = {}
		for b, (_, _, idxs) in enumerate(self.buckets):
			for i in idxs:
				self.bucket_of[i] = b

	def zero_grad(self, fill=True):
		if fill:
			self.grad.zero_()
		# autograd replaces .grad when it is None; keep the views attached
		for p, o in zip(self.params, self.offsets):
			if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
				p.grad = self.grad[o:o + p.numel()].view(p.shape)


def broadcast_replica(flat, src=0, process_group=None, single_rank=False):
	"""Every rank starts from rank `src`'s replica: ONE broadcast of the flat parameter buffer (the trainable parameters
	are views into it), then the few tensors outside it (frozen parameters, BatchNorm statistics).  No-op in a single
	process (single_rank: run the collectives in a process group of one rank as well, see GradSync)."""
	if not (dist.is_initialized() and (dist.get_world_size(process_group) > 1 or single_rank)):
		return
	dist.broadcast(flat.param, src=src, group=process_group)
	inside = {id(p) for p in flat.params}
	for p in flat.module.parameters():
		if id(p) not in inside:
			dist.broadcast(p.data, src=src, group=process_group)
	for b in flat.module.buffers():
		dist.broadcast(b.data, src=src, group=process_group)


def _cast_into(src, dst):
	"""dst <- src converted to dst's dtype (float32 <-> bfloat16), same element count: vitk_cast on the GPU, a plain copy on
	the CPU (the gloo tests)."""
	if src.is_cuda:
		L.check(L.load().vitk_cast(L.ptr(src), L.ptr(dst), src.numel(), L.dtype_code(src.dtype), L.dtype_code(dst.dtype), L.stream()))
	else:
		dst.copy_(src)


class GradSync:
	"""Bucketed all-reduce(sum) of FlatModel.grad, launched from autograd hooks during backward.

	A gradient may be written several times per step (a module called twice, tied weights, several sink-aware backward
	functions on one parameter), so a parameter counts as final only after as many reports as its FIRST synchronised
	backward produced; that first backward launches nothing early (all buckets go out in finish()).  Buckets are always
	launched in index order (the order their gradients become final in a backward pass), each as soon as it and all its
	predecessors are complete, so every rank issues the same sequence of collectives even when the ranks' sets of
	unused parameters differ.  Inside no_sync() backward passes only accumulate (micro-batches); the reports of the
	next backward outside it are counted."""

	def __init__(self, flat, process_group=None, single_rank=False, bf16_buckets=False):
		"""single_rank: keep the whole machinery (hooks, bucket launches, waits) active in an initialised process group of
		ONE rank, where every collective is the identity: the RCCL call sequence of a multi-GPU step, runnable on one GPU.
		bf16_buckets: a bucket travels as bfloat16 (rounded into a staging buffer of the same layout when it is launched,
		summed by the collective in bfloat16, widened back into the float32 gradient in finish()): half the bytes on the
		xGMI links (173 instead of 346 MB per ViT-B/16 step) for a gradient that is exact to 2^-9 per rank; off by default."""
		self.flat = flat
		self.bf16_buckets = bool(bf16_buckets)
		self.stage = torch.empty(flat.numel, dtype=torch.bfloat16, device=flat.grad.device) if self.bf16_buckets else None
		self.group = process_group
		self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
		self.active = self.world > 1 or (single_rank and dist.is_initialized())
		self.expected = None  # reports per parameter in one backward, learnt from the first synchronised one
		self.enabled = True
		self.works = []
		self.launched = []
		self.hooks = [None] * len(flat.params)
		self._cus_free_before = None
		if self.active:
			if flat.param.is_cuda:
				# the persistent GEMMs leave the CUs they cannot fill a round with to the communication kernels (a process-wide
				# option of the library: close() puts the previous value back)
				self._cus_free_before = L.load().vitk_gemm_leave_cus_free(1)
			for i, p in enumerate(flat.params):
				self.hooks[i] = self._make_hook(i)
				p.register_post_accumulate_grad_hook(self.hooks[i])
		self.reset()

	def reset(self):
		self.counts = [0] * len(self.flat.params)
		self.works = []
		self.launched = []
		self.arrivals = []  # parameter indices in the order their gradients became final (diagnostics / tests)
		self.next_bucket = 0
		self.late = None
		if self.expected is not None:
			self.pending = [sum(1 for i in idxs if self.expected[i] > 0) for _, _, idxs in self.flat.buckets]
		else:
			self.pending = [len(idxs) for _, _, idxs in self.flat.buckets]

	class _NoSync:
		def __init__(self, sync):
			self.sync = sync

		def __enter__(self):
			self.prev, self.sync.enabled = self.sync.enabled, False

		def __exit__(self, *exc):
			self.sync.enabled = self.prev

	def no_sync(self):
		"""Context manager: backward passes inside it accumulate into the flat gradient without any collective (gradient
		accumulation over micro-batches); run the last micro-batch's backward outside it."""
		return GradSync._NoSync(self)

	def _make_hook(self, i):
		def hook(_param):
			if not self.enabled:
				return
			self.counts[i] += 1
			if self.expected is None:
				return  # first synchronised backward: learn the counts, launch everything in finish()
			b = self.flat.bucket_of[i]
			if self.counts[i] == self.expected[i]:
				self.arrivals.append(i)
				self.pending[b] -= 1
				self._launch_ready()
			elif self.counts[i] > self.expected[i] and b < self.next_bucket:
				self.late = i  # written after its bucket went out: the replicas would diverge; finish() raises
		return hook

	def _launch_ready(self):
		while self.next_bucket < len(self.flat.buckets) and self.pending[self.next_bucket] <= 0:
			self._launch(self.next_bucket)
			self.next_bucket += 1

	def _launch(self, b):
		"""A collective is ordered behind the stream that is current where it is launched.  Weight-gradient products of the
		backward pass may still be adding into this bucket on the second stream (ops.linear_wgrad(overlap=True); joined only
		at the end of backward), and a hook that autograd fires itself -- a parameter whose gradient came back through
		autograd's own accumulation: LayerScale, conv-embedding, MBConv, DyT parameters -- runs with the main stream current:
		the launch then goes behind BOTH streams (arrivals reported by the sink-aware backward functions come here that way
		already, through ops._sink_done)."""
		start, end, _ = self.flat.buckets[b]
		self.launched.append(b)
		grad = self.flat.grad
		lane = ops.lane_in_flight(grad.device) if grad.is_cuda else None
		if lane is not None and L.stream() != lane.stream.cuda_stream:
			lane.run_behind(self._all_reduce, start, end)
		else:
			self._all_reduce(start, end)

	def _all_reduce(self, start, end):
		buf = self.flat.grad[start:end]
		if self.bf16_buckets:
			buf = self.stage[start:end]
			_cast_into(self.flat.grad[start:end], buf)  # on the stream the collective is ordered behind
		self.works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

	def close(self):
		"""Put the library's process-wide grid option back to what it was before this object set it."""
		if self._cus_free_before is not None:
			L.load().vitk_gemm_leave_cus_free(self._cus_free_before)
			self._cus_free_before = None

	def finish(self):
		"""Wait for the collectives of this step (launching, in order, every bucket that has not gone out: the first
		step's, and those holding parameters that received fewer gradient writes than expected, e.g. unused ones).
		After this the flat gradient holds the SUM over ranks; the optimizer applies 1/world."""
		if self.active:
			if self.late is not None:
				name = self.flat.names[self.late]
				self.reset()
				raise RuntimeError(f"gradient of {name} was written after its bucket had been all-reduced (the autograd graph changed between steps); "
					"build a new GradSync / Trainer for the new graph")
			while self.next_bucket < len(self.flat.buckets):
				self._launch(self.next_bucket)
				self.next_bucket += 1
			for w in self.works:
				w.wait()
			if self.bf16_buckets:  # every bucket has been launched by now: the sums go back into the float32 gradient
				_cast_into(self.stage, self.flat.grad)
			if self.expected is None and any(self.counts):  # (a step whose hooks were silenced teaches nothing)
				self.expected = list(self.counts)
		self.reset()


class FlatAdamW(torch.optim.Optimizer):
	"""torch.optim.AdamW semantics (main.py:100 there) as ONE kernel over the flat buffers, which also writes the
	bfloat16 copy of the parameters that the next forward's GEMMs read.

	It is a torch.optim.Optimizer with the reference's two parameter groups (utils.py:102-121 there: decayed parameters,
	then the undecayed ones), so the stock learning-rate schedulers and the reference's warm-up wrapper
	(schedulers.py:3-35, main.py:113-131) attach to it and drive `param_groups[i]["lr"]`.  The kernel reads the
	hyper-parameters of both groups from a small device buffer that step() refreshes from `param_groups` whenever they
	changed, so a step captured into a HIP graph follows the schedule as well (GraphedStep refreshes the buffer before
	each replay)."""

	def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, decay_filter=no_decay, bf16_shadow=True):
		self.flat = flat
		dec = [flat.params[i] for i in flat.module_order if not decay_filter(flat.names[i])]
		nod = [flat.params[i] for i in flat.module_order if decay_filter(flat.names[i])]
		groups = [{"params": dec, "weight_decay": weight_decay}, {"params": nod, "weight_decay": 0.0}]
		groups = [g for g in groups if g["params"]] or groups[:1]
		super().__init__(groups, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
		self._decay_group = 0 if dec else None
		self._nodecay_group = (1 if dec else 0) if nod else None
		dev = flat.param.device
		self.m = torch.zeros_like(flat.param)
		self.v = torch.zeros_like(flat.param)
		self.step_count = 0
		self.step_state = torch.zeros(2, dtype=torch.int64, device=dev)
		self.hparams = torch.zeros(8, dtype=torch.float32, device=dev)
		self._hparams_host = torch.zeros(8, dtype=torch.float32).pin_memory() if dev.type == "cuda" else torch.zeros(8)
		self._hparams_sent = None
		mask = torch.ones(flat.numel, dtype=torch.uint8, device="cpu")
		for n, p, o in zip(flat.names, flat.params, flat.offsets):
			if decay_filter(n):
				mask[o:o + p.numel()] = 0
		self.wd_mask = mask.to(dev)
		self.shadow = None
		self._registered = []
		if bf16_shadow:
			self.shadow = torch.empty(flat.numel, dtype=torch.bfloat16, device=dev)
			L.check(L.load().vitk_cast(L.ptr(flat.param), L.ptr(self.shadow), flat.numel, L.F32, L.BF16, L.stream()))
			# transposed bf16 copies of the 2-D weights (the input-gradient GEMMs read them), refreshed by ONE launch per step
			self.shadow_t = torch.empty(flat.numel, dtype=torch.bfloat16, device=dev)
			items, tiles = [], []
			self._tile_range = {}  # parameter index -> [first, end) in the tile list of the batched transpose
			for pi, (p, o) in enumerate(zip(flat.params, flat.offsets)):
				if p.dim() >= 2:
					sh = self.shadow[o:o + p.numel()].view(p.shape)
					ops.register_shadow(p, sh, self.refresh_copies)
					self._registered.append((p, sh))
					if p.dim() == 2:
						rows, cols = p.shape
						sht = self.shadow_t[o:o + p.numel()].view(cols, rows)
						ops.register_transposed(sh, sht)
						first_tile = len(tiles)
						for tr in range((rows + 63) // 64):
							for tc in range((cols + 63) // 64):
								tiles.append((len(items), tr, tc, 0))
						items.append((sh.data_ptr(), sht.data_ptr(), rows, cols))
						self._tile_range[pi] = (first_tile, len(tiles))
			self.t_items = torch.tensor(items, dtype=torch.int64).reshape(-1, 4).to(dev)  # vitk_transpose_item: 2 pointers + 2 int64
			self.t_tiles = torch.tensor(tiles, dtype=torch.int32).reshape(-1, 4).to(dev)
			self._refresh_transposed()
		self.sync_hparams()
		# pieces of the flat buffers for the pipelined step: one per top-level child of the module (per transformer block
		# inside a ModuleList), in buffer order = the order a forward pass first reads them
		self.pieces = []  # [start, end, first tile, end tile, parameter indices]
		key_of = lambda n: ".".join(n.split(".")[:2]) if n.split(".")[0] in ("blocks", "layers", "stages") else n.split(".")[0]
		prev = None
		for pi, (n, p, o) in enumerate(zip(flat.names, flat.params, flat.offsets)):
			k = key_of(n)
			if k != prev:
				self.pieces.append([o, o, None, None, []])
				prev = k
			pc = self.pieces[-1]
			pc[1] = o + (p.numel() + 63) // 64 * 64
			pc[4].append(pi)
			tr = getattr(self, "_tile_range", {}).get(pi)
			if tr is not None:
				pc[2] = tr[0] if pc[2] is None else pc[2]
				pc[3] = tr[1]
		assert self.pieces[0][0] == 0 and self.pieces[-1][1] == flat.numel and all(a[1] == b[0] for a, b in zip(self.pieces, self.pieces[1:]))
		self.grads_cleared = False  # the last step cleared the gradients it consumed (pipelined step)

	# the first group's hyper-parameters under the names the fused optimizer had before it grew parameter groups
	def _g0(self):
		return self.param_groups[0]

	lr = property(lambda self: self._g0()["lr"], lambda self, v: [g.__setitem__("lr", v) for g in self.param_groups] and None)
	betas = property(lambda self: self._g0()["betas"], lambda self, v: [g.__setitem__("betas", tuple(v)) for g in self.param_groups] and None)
	eps = property(lambda self: self._g0()["eps"], lambda self, v: [g.__setitem__("eps", v) for g in self.param_groups] and None)

	@property
	def weight_decay(self):
		return self.param_groups[self._decay_group]["weight_decay"] if self._decay_group is not None else 0.0

	@weight_decay.setter
	def weight_decay(self, v):
		if self._decay_group is not None:
			self.param_groups[self._decay_group]["weight_decay"] = v

	def _hparam_values(self):
		g0 = self.param_groups[self._decay_group] if self._decay_group is not None else None
		g1 = self.param_groups[self._nodecay_group] if self._nodecay_group is not None else None
		ref = g0 or g1
		b1, b2 = ref["betas"]
		f = lambda g, k: float(g[k]) if g is not None else 0.0
		return (f(g0, "lr"), f(g0, "weight_decay"), f(g1, "lr"), f(g1, "weight_decay"), float(b1), float(b2), float(ref["eps"]), 0.0)

	def sync_hparams(self):
		"""Copy the hyper-parameters of `param_groups` to the device buffer the kernel reads, if they changed (never
		during stream capture: a captured step picks them up from the buffer)."""
		vals = self._hparam_values()
		if vals == self._hparams_sent or (self.hparams.is_cuda and torch.cuda.is_current_stream_capturing()):
			return
		self._hparams_host.copy_(torch.tensor(vals, dtype=torch.float32))
		self.hparams.copy_(self._hparams_host, non_blocking=False)
		self._hparams_sent = vals

	def _refresh_transposed(self):
		if self.shadow is not None and self.t_tiles.shape[0] > 0:
			L.check(L.load().vitk_transpose_batched(L.ptr(self.t_items), L.ptr(self.t_tiles), self.t_tiles.shape[0], L.BF16, L.stream()))

	def refresh_copies(self):
		"""Re-derive the bfloat16 and transposed copies after the float32 parameters were written from outside
		(load_state_dict, a broadcast).  weight_as() calls this by itself when a parameter's version counter moved."""
		self.wait_pipelined()
		if self.shadow is not None:
			L.check(L.load().vitk_cast(L.ptr(self.flat.param), L.ptr(self.shadow), self.flat.numel, L.F32, L.BF16, L.stream()))
			self._refresh_transposed()
			for p, _ in self._registered:
				ops.shadow_in_sync(p)
		ops.invalidate_casts()
		ops.weights_changed()

	def close(self):
		"""Withdraw the copies registered with ops (they would otherwise outlive this optimizer with stale contents)."""
		self.wait_pipelined()
		for p, sh in self._registered:
			ops.unregister_shadow(p)
			ops.unregister_transposed(sh)
		self._registered = []

	def set_step(self, n):
		"""Set the number of steps taken so far (resuming from a checkpoint)."""
		self.wait_pipelined()
		self.step_count = int(n)
		self.step_state.fill_(int(n))

	def zero_grad(self, set_to_none=False):
		self.flat.zero_grad(fill=not self.grads_cleared)  # the gradients stay views of the flat buffer
		self.grads_cleared = False

	def state_dict(self):
		from . import checkpoint
		return checkpoint.optimizer_state_dict(self)

	def load_state_dict(self, sd):
		from . import checkpoint
		checkpoint.load_optimizer_state_dict(self, sd)

	def step(self, closure=None, grad_scale=1.0, pipelined=False):
		"""pipelined: issue the step on the second stream, piece by piece in the order the next forward pass
		reads the parameters, clearing each gradient as it is consumed (no zero_grad pass), and return at once: every
		kernel of this package that reads a parameter first makes its stream wait for that parameter's piece
		(ops.weight_as), so the next forward starts while the later blocks are still being updated, and the update runs
		on the CUs its GEMMs leave idle.  Anything ELSE that reads parameters, moments or gradients (a checkpoint, a
		comparison in a test, .cpu()) must call wait_pipelined() first.  Ignored under HIP graph capture."""
		# the step count that the bias corrections use lives in device memory (self.step_state) and is advanced by the
		# launch itself, so the same launch can be replayed from a captured HIP graph; step_count mirrors it on the host
		assert closure is None, "the fused step takes no closure"
		self.step_count += 1
		self.sync_hparams()
		f = self.flat
		if pipelined and self.shadow is not None and f.param.is_cuda and not torch.cuda.is_current_stream_capturing():
			return self._step_pipelined(grad_scale)
		self.grads_cleared = False  # (this form leaves the gradients as they are)
		L.check(L.load().vitk_adamw_device_hparams(L.ptr(f.param), L.ptr(f.grad), L.ptr(self.m), L.ptr(self.v), L.ptr(self.wd_mask), L.ptr(self.shadow), f.numel,
			L.ptr(self.hparams), L.ptr(self.step_state), grad_scale, L.stream()))
		if self.shadow is None:
			ops.invalidate_casts()  # the kernel writes the parameters without bumping their version counters
		self._refresh_transposed()
		ops.weights_changed()  # transposed copies cached by ops itself (weights without a kept copy) are stale now


	def _step_pipelined(self, grad_scale):
		f, lib = self.flat, L.load()
		lane = ops.second_stream(f.param.device)
		lane.stream.wait_stream(torch.cuda.current_stream())  # the gradients are complete (backward has joined, collectives were waited for)
		pending = []
		with torch.cuda.stream(lane.stream):
			L.check(lib.vitk_adamw_advance(L.ptr(self.step_state), L.ptr(self.hparams), L.stream()))
			for start, end, t0, t1, idxs in self.pieces:
				off = lambda t, esz: t.data_ptr() + start * esz
				L.check(lib.vitk_adamw_range(off(f.param, 4), off(f.grad, 4), off(self.m, 4), off(self.v, 4), off(self.wd_mask, 1), off(self.shadow, 2), end - start,
					L.ptr(self.hparams), L.ptr(self.step_state), grad_scale, 1, L.stream()))
				if t0 is not None and t1 > t0:
					L.check(lib.vitk_transpose_batched(L.ptr(self.t_items), self.t_tiles.data_ptr() + 16 * t0, t1 - t0, L.BF16, L.stream()))
				ev = torch.cuda.Event()
				ev.record()
				pending.append((ev, [f.params[i] for i in idxs]))
		ops.set_pending_weights(pending, lane)
		self.grads_cleared = True
		ops.weights_changed()

	def wait_pipelined(self):
		"""The current stream waits for a pipelined step still in flight (a no-op otherwise)."""
		ops.wait_pending_weights()


class Trainer:
	"""zero_grad -> forward/backward (caller) -> step, with gradient sync when torch.distributed is initialised."""

	def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, bucket_bytes=48 << 20, single_rank_sync=False, pipelined_step=False,
			bf16_buckets=False):
		"""pipelined_step: step() returns while the optimizer still runs on the second stream (FlatAdamW.step); the next
		forward / backward of this package synchronise with it by themselves, other readers call wait_step() first.
		bf16_buckets: gradient buckets are all-reduced as bfloat16 (GradSync)."""
		self.pipelined_step = pipelined_step
		self.flat = FlatModel(module, bucket_bytes)
		broadcast_replica(self.flat, single_rank=single_rank_sync)
		self.sync = GradSync(self.flat, single_rank=single_rank_sync, bf16_buckets=bf16_buckets)
		self.opt = FlatAdamW(self.flat, lr, betas, eps, weight_decay)
		# backward kernels that know about gradient sinks add straight into the flat gradient buffer (no temporary, no
		# autograd accumulation kernel); they report through the same arrival hook as autograd's accumulation
		for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
			ops.register_grad_sink(p, self.flat.grad[o:o + p.numel()].view(p.shape), self.sync.hooks[i])

	def zero_grad(self):
		if self.opt.grads_cleared:  # the pipelined step cleared every gradient as it consumed it
			self.opt.grads_cleared = False
			self.flat.zero_grad(fill=False)
		else:
			self.flat.zero_grad()

	def wait_step(self):
		"""The current stream waits for a pipelined optimizer step still in flight."""
		self.opt.wait_pipelined()

	def no_sync(self):
		"""Gradient accumulation: `with trainer.no_sync(): loss.backward()` for all micro-batches but the last."""
		return self.sync.no_sync()

	def step(self):
		self.sync.finish()
		self.opt.step(grad_scale=1.0 / self.sync.world, pipelined=self.pipelined_step)

	def close(self):
		"""Withdraw everything this trainer registered with ops (bf16 / transposed weight copies, gradient sinks)."""
		self.opt.close()
		self.sync.close()
		for p in self.flat.params:
			ops.unregister_grad_sink(p)


class GraphedStep:
	"""One whole training step -- zero_grad, forward + loss (`loss_fn`), backward, optimizer -- captured once into a HIP
	graph and replayed per call: one graph launch instead of ~10^3 kernel launches, which is what a small model
	(ViT-Tiny on 32x32: ~1 ms of GPU work per step behind ~8 ms of launch overhead) is bound by.  The reference's loop
	(train.py:42-59) is eager and synchronises twice per step; this is the launch-free form of the same step.

	`loss_fn(*inputs)` returns the scalar loss, or (loss, logits), and may only depend on its tensor arguments (their
	shapes and dtypes are fixed at capture; new values are copied into the captured input buffers).  With more than one rank
	the graph holds forward + backward only and the gradient all-reduces and the optimizer step follow it eagerly.  The warm-up steps that precede the capture (lazy one-time set-up
	must not happen while capturing) run on the example inputs and are then UNDONE: parameters, optimizer moments, step
	count, module buffers (BatchNorm statistics) and the dropout seed counter are restored, so the first replay is the
	first training step and the trajectory equals the eager one.  Learning rate and the other hyper-parameters are
	re-read from the optimizer's param_groups before every replay.  Dropout seeds are host-side values fixed at capture;
	the captured step starts by advancing a device word that every mask mixes into its seed (ops.DropoutEpoch), so each
	replay draws new masks (with dropout > 0 the trajectory equals the eager one in distribution, not mask for mask).  The
	word stays registered on the device until close().  The returned tensors are overwritten by the next call."""

	def __init__(self, trainer, loss_fn, example_inputs, warmup=3):
		# With gradient synchronisation active (more than one rank) the collectives stay OUTSIDE the graph: the captured part is
		# zero_grad + forward + backward with the hooks silenced (GradSync.no_sync), and every call then launches the buckets'
		# all-reduces and the optimizer step eagerly behind the replay -- a handful of launches instead of ~10^3, which is what
		# the launch-bound 32x32 configurations need on every rank of a node (a ViT-Tiny / CvT gradient is 21 / 80 MB: the
		# collectives cannot hide under a 1 ms backward anyway).
		self.sharded = trainer.sync.active
		# a captured step is the plain sequence zero_grad -> forward -> backward -> optimizer on one stream: a pipelined step
		# cannot be captured, and a warm-up step that cleared the gradients itself would leave the capture without its zero fill
		trainer.wait_step()
		trainer.pipelined_step = False
		trainer.opt.grads_cleared = False
		assert ops.GEMM_PROFILE is None, "no per-launch event timing inside a captured step"
		self.trainer, self.loss_fn = trainer, loss_fn
		self.inputs = [t.clone() for t in example_inputs]
		opt, flat = trainer.opt, trainer.flat
		saved = [t.clone() for t in (flat.param, opt.m, opt.v, opt.step_state)]
		bufs = [(b, b.clone()) for b in flat.module.buffers()]
		saved_steps, saved_seed = opt.step_count, ops.seed_counter()
		self.epoch = ops.DropoutEpoch(flat.param.device)
		side = torch.cuda.Stream()
		side.wait_stream(torch.cuda.current_stream())
		with torch.cuda.stream(side):  # warm-up off the default stream
			for _ in range(warmup):
				self._eager()
		torch.cuda.current_stream().wait_stream(side)
		self.graph = torch.cuda.CUDAGraph()
		with torch.cuda.graph(self.graph):
			out = self._eager()
		self.loss, self.logits = out if isinstance(out, tuple) else (out, None)
		for dst, src in zip((flat.param, opt.m, opt.v, opt.step_state), saved):
			dst.copy_(src)
		for b, src in bufs:
			b.copy_(src)
		opt.step_count = saved_steps
		ops.set_seed_counter(saved_seed)
		self.epoch.set(0)
		opt.refresh_copies()

	def close(self):
		"""Unregister the dropout-epoch word (the captured graph must not be replayed afterwards)."""
		self.epoch.close()

	def _eager(self):
		self.epoch.advance()
		self.trainer.zero_grad()
		out = self.loss_fn(*self.inputs)
		loss, logits = out if isinstance(out, tuple) else (out, None)
		if self.sharded:
			with self.trainer.no_sync():  # the hooks launch nothing: the collectives are issued by __call__, outside the graph
				loss.backward()
			if not torch.cuda.is_current_stream_capturing():
				self.trainer.step()  # (warm-up steps: the eager tail, as every call runs it)
		else:
			loss.backward()
			self.trainer.step()
		return loss.detach() if logits is None else (loss.detach(), logits.detach())

	def __call__(self, *inputs):
		for dst, src in zip(self.inputs, inputs):
			if dst.data_ptr() != src.data_ptr():
				dst.copy_(src, non_blocking=True)
		self.trainer.opt.sync_hparams()
		self.graph.replay()
		if self.sharded:
			self.trainer.step()  # every bucket's all-reduce (in order), the waits, the fused optimizer step: eager
		else:
			self.trainer.opt.step_count += 1
		return self.loss if self.logits is None else (self.loss, self.logits)
