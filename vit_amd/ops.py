"""Tensor-level wrappers over the vitk C ABI and the autograd Functions built from them.

Layout conventions: activations are row-major [rows, features] (any leading dims flattened),
weights keep the reference's nn.Linear layout [out_features, in_features].  `compute dtype` is the
dtype of the activations (float32 or bfloat16); weights are cast to it through a version-keyed
cache, parameter gradients are produced in float32 and cast to the parameter dtype if needed.
"""
import contextlib
import math
import os
import weakref

import torch

from . import _lib as L


# --------------------------------------------------------------------------------------------------
# dtype plumbing
# --------------------------------------------------------------------------------------------------

def compute_dtype(x):
	"""bfloat16 when the input is bfloat16 or bf16 autocast is active, else float32."""
	if x.dtype == torch.bfloat16:
		return torch.bfloat16
	if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
		return torch.bfloat16
	if x.dtype == torch.float32:
		return torch.float32
	raise TypeError(f"unsupported activation dtype {x.dtype}")


def cast(t, dtype):
	"""Contiguous copy of t in `dtype` through vitk_cast (no-op if already there)."""
	if t.dtype == dtype and t.is_contiguous():
		return t
	if not t.is_contiguous():
		t = t.contiguous()
	if t.dtype == dtype:
		return t
	out = torch.empty(t.shape, dtype=dtype, device=t.device)
	L.check(L.load().vitk_cast(L.ptr(t), L.ptr(out), t.numel(), L.dtype_code(t.dtype), L.dtype_code(dtype), L.stream()))
	return out


_cast_cache = {}  # id(param) -> (weakref(param), key, cast copy)


_shadows = {}  # id(param) -> [weakref(param), bf16 copy kept in sync by the optimizer kernel, param._version it matches, refresh callback]


def register_shadow(param, shadow_bf16, refresh=None):
	"""Declare `shadow_bf16` to be a bfloat16 copy of `param` that its owner keeps up to date (the fused AdamW
	kernel rewrites it with every step, without touching the parameter's version counter); weight_as() then returns it
	without casting.  The parameter's version is recorded: when anything else writes the parameter in place
	(load_state_dict, an initialiser, another optimizer) the version moves, and weight_as() calls `refresh()` — which must
	re-derive the copies and call shadow_in_sync() — or raises if there is none, instead of computing with stale weights."""
	assert shadow_bf16.dtype == torch.bfloat16 and shadow_bf16.shape == param.shape and shadow_bf16.is_contiguous()
	pid = id(param)
	_shadows[pid] = [weakref.ref(param, lambda _r, pid=pid, d=_shadows: d.pop(pid, None)), shadow_bf16, param._version, refresh]


def shadow_in_sync(param):
	"""The owner of `param`'s registered copy has just rewritten it from the parameter's current value."""
	rec = _shadows.get(id(param))
	if rec is not None and rec[0]() is param:
		rec[2] = param._version


def unregister_shadow(param):
	_shadows.pop(id(param), None)


def invalidate_casts():
	"""Drop cached casts (call after modifying parameters through anything that does not bump their version)."""
	_cast_cache.clear()
	_wt_cache.clear()


# Parameters that a pipelined optimizer step (dp.FlatAdamW.step(pipelined=True)) is still updating on the second stream:
# id(param) -> [event behind the update of its piece, handles of the streams that wait for it already].  weight_as() is the
# one door through which the kernels of this package reach a parameter (its float32 master, its bf16 copy, the transposed
# copy behind that), so the wait lives there; one wait per piece AND stream (a marker packet on the stream costs a few
# microseconds; a reader on another stream -- an evaluation or teacher forward on a side stream, a launch made while the
# second stream is current -- must wait for itself).
_pending_weights = {}
_pending_lane = None


def set_pending_weights(pieces, lane):
	"""pieces: [(event, [parameters])] of a step just issued on `lane`."""
	global _pending_lane
	for k in [k for k, rec in _pending_weights.items() if rec[1]]:  # (waited for already; records of another optimizer stay)
		del _pending_weights[k]
	for ev, params in pieces:
		rec = [ev, set()]
		for p in params:
			_pending_weights[id(p)] = rec
	_pending_lane = lane


def wait_pending_weights():
	"""The current stream waits for the whole pipelined step in flight, if any."""
	global _pending_lane
	if _pending_lane is not None:
		torch.cuda.current_stream().wait_stream(_pending_lane.stream)
		_pending_lane = None
	_pending_weights.clear()


def _await_weight(w):
	if _pending_weights:
		rec = _pending_weights.get(id(w))
		if rec is not None:
			sid = L.stream()
			if sid not in rec[1]:
				torch.cuda.current_stream().wait_event(rec[0])
				rec[1].add(sid)


def weight_as(w, dtype):
	"""`w` in the compute dtype.  The cast of a parameter is cached until the parameter is modified
	in place (optimizer step, load_state_dict), which bumps its version counter."""
	_await_weight(w)
	if w.dtype == dtype and w.is_contiguous():
		return w.detach()
	if dtype == torch.bfloat16:
		sh = _shadows.get(id(w))
		if sh is not None and sh[0]() is w:
			if sh[2] != w._version:
				if sh[3] is None:
					raise RuntimeError("a parameter with a registered bfloat16 copy was modified in place by something other than its optimizer; "
						"call the optimizer's refresh_copies() after writing parameters")
				sh[3]()
			return sh[1]
	key = (w._version, dtype, w.data_ptr())
	hit = _cast_cache.get(id(w))
	if hit is not None and hit[0]() is w and hit[1] == key:
		return hit[2]
	out = cast(w.detach(), dtype)
	wid = id(w)
	_cast_cache[wid] = (weakref.ref(w, lambda _r, wid=wid, d=_cast_cache: d.pop(wid, None)), key, out)
	return out


def weight_padded(w, dtype, rows, cols, ld):
	"""The parameter `w` viewed as [rows, cols], in `dtype`, with rows of `ld` >= cols elements (zeros behind the data): one
	kernel, cached until the parameter is modified in place."""
	if ld == cols:
		return weight_as(w, dtype).reshape(rows, cols)
	_await_weight(w)
	# (the fused optimizer rewrites the parameter without bumping its version counter: its epoch is part of the key)
	key = (w._version, dtype, w.data_ptr(), ld, _weights_epoch)
	hit = _cast_cache.get((id(w), "pad"))
	if hit is not None and hit[0]() is w and hit[1] == key:
		return hit[2]
	src = w.detach()
	assert src.is_contiguous() and src.numel() == rows * cols
	out = torch.empty((rows, ld), dtype=dtype, device=w.device)
	L.check(L.load().vitk_pad_cols(L.ptr(src), L.ptr(out), rows, cols, cols, ld, L.dtype_code(src.dtype), L.dtype_code(dtype), L.stream()))
	k = (id(w), "pad")
	_cast_cache[k] = (weakref.ref(w, lambda _r, k=k, d=_cast_cache: d.pop(k, None)), key, out)
	return out


def transpose(x):
	"""Contiguous transpose of a 2-D tensor through vitk_transpose."""
	assert x.dim() == 2 and x.stride(1) == 1
	rows, cols = x.shape
	y = torch.empty((cols, rows), dtype=x.dtype, device=x.device)
	L.check(L.load().vitk_transpose(L.ptr(x), L.ptr(y), rows, cols, x.stride(0), rows, L.dtype_code(x.dtype), L.stream()))
	return y


_wt_cache = {}  # id(compute-dtype weight) -> (weakref, key, transposed copy)
_weights_epoch = 0


def weights_changed():
	"""Tell the caches that weight copies were rewritten behind torch's back (the fused optimizer kernel updates the
	bf16 shadows in place without bumping a version counter)."""
	global _weights_epoch
	_weights_epoch += 1


_wt_kept = {}  # id(compute-dtype weight) -> (weakref, transposed copy that its owner keeps up to date)


def register_transposed(wc, wt):
	"""Declare `wt` [in, out] to be a transposed copy of the compute-dtype weight `wc` [out, in] (e.g. a registered
	bf16 shadow) that the owner refreshes whenever the weight changes (the optimizer does, with one batched launch)."""
	assert wt.shape == (wc.shape[1], wc.shape[0]) and wt.dtype == wc.dtype and wt.is_contiguous()
	wid = id(wc)
	_wt_kept[wid] = (weakref.ref(wc, lambda _r, wid=wid, d=_wt_kept: d.pop(wid, None)), wt)


def weight_t(wc):
	"""Transposed copy [in, out] of a weight `wc` [out, in] that is already in the compute dtype (what weight_as
	returned); cached until the weight changes."""
	kept = _wt_kept.get(id(wc))
	if kept is not None and kept[0]() is wc:
		return kept[1]
	key = (wc._version, wc.data_ptr(), _weights_epoch)
	hit = _wt_cache.get(id(wc))
	if hit is not None and hit[0]() is wc and hit[1] == key:
		return hit[2]
	out = transpose(wc)
	wid = id(wc)
	_wt_cache[wid] = (weakref.ref(wc, lambda _r, wid=wid, d=_wt_cache: d.pop(wid, None)), key, out)
	return out


_grad_sinks = {}  # id(param) -> (weakref(param), float32 buffer of the parameter's shape, callback or None)


def register_grad_sink(param, buffer, on_written=None):
	"""Declare `buffer` (float32, the parameter's shape, contiguous; typically a slice of a flat gradient buffer that
	param.grad is a view of) as the place where backward kernels ACCUMULATE the gradient of `param` directly: the
	sink-aware backward functions then add into it (weight-gradient GEMM with accumulate, bias / norm gradients by the
	kernels' own accumulation) and hand autograd `None` for that parameter, which saves the gradient temporary, its
	zero-fill and autograd's `grad += temporary` kernel.  `on_written(param)` is called once per backward of a
	function that wrote the sink (the data-parallel trainer counts these like post-accumulate-grad hooks).  The
	owner zeroes the buffer between steps.  A parameter must not also receive a gradient through plain autograd in the
	same backward if the callback counts arrivals."""
	assert buffer.dtype == torch.float32 and buffer.shape == param.shape and buffer.is_contiguous()
	pid = id(param)
	_grad_sinks[pid] = (weakref.ref(param, lambda _r, pid=pid, d=_grad_sinks: d.pop(pid, None)), buffer, on_written)


def clear_grad_sinks():
	_grad_sinks.clear()


def unregister_grad_sink(param):
	_grad_sinks.pop(id(param), None)


def unregister_transposed(wc):
	_wt_kept.pop(id(wc), None)


def _sink(p):
	"""(buffer, callback) if gradients of parameter `p` go to a registered sink, else None."""
	if p is None:
		return None
	s = _grad_sinks.get(id(p))
	if s is None or s[0]() is not p:
		return None
	return s[1], s[2]


def _sink_done(p, sink):
	"""Autograd's return value for a parameter whose gradient was accumulated into its sink."""
	if sink[1] is not None and _wgrad_held is not None:
		_wgrad_held_done.append((p, sink))  # products into sinks are still held back: report behind their launch (wgrad_group)
		return None
	if sink[1] is not None:
		lane = _wgrad_lanes.get(p.device.index) if p.is_cuda else None
		if lane is not None and (lane.held or lane.hooked):
			lane.run_behind(sink[1], p)  # (the callback may launch the bucket's all-reduce: behind the weight-gradient stream too)
		else:
			sink[1](p)
	return None


class _Sinks:
	"""The gradient sinks of one backward function's parameters: `buf(name)` is the buffer a kernel should add into (or
	None), `ret(name, grad)` is what the function hands autograd for that parameter."""

	def __init__(self, **params):
		self.params = params
		self.sinks = {n: _sink(p) for n, p in params.items()}

	def buf(self, name):
		s = self.sinks[name]
		return s[0] if s is not None else None

	def ret(self, name, grad):
		p, s = self.params[name], self.sinks[name]
		if p is None:
			return None
		return _sink_done(p, s) if s is not None else grad_as(grad, p)


def f32(t):
	"""float32 view/copy of a small parameter vector (norm weights, biases, LayerScale)."""
	if t is None:
		return None
	return weight_as(t, torch.float32)


def grad_as(g, like):
	if g is None or like is None:
		return None
	return g if g.dtype == like.dtype else cast(g, like.dtype)


# --------------------------------------------------------------------------------------------------
# raw kernels
# --------------------------------------------------------------------------------------------------

DGRAD_TRANSPOSED_WEIGHTS = True  # see linear_dgrad
DGRAD_T_MIN = 64  # smallest layer width that takes the transposed-copy path (ViT-Tiny, d = 192: +3.5 % on the step)
GEMM_PROFILE = None  # set to a list to collect (kernel label, flops, start event, end event) per launch
GEMM_PROFILE_SERIAL = False  # keep every GEMM on one stream (per-kernel timing without another kernel sharing the CUs)


_workspaces = {}  # (device index, on the weight-gradient stream) -> uint8 scratch tensor handed to vitk_gemm (K-split partial tiles)


def gemm_workspace(device):
	key = (device.index, _on_wgrad_stream)
	ws = _workspaces.get(key)
	if ws is None:
		ws = torch.empty(int(L.load().vitk_gemm_workspace_bytes()), dtype=torch.uint8, device=device)
		_workspaces[key] = ws
	return ws


# --------------------------------------------------------------------------------------------------
# weight gradients on a second stream
# --------------------------------------------------------------------------------------------------
# dW = dy^T x of a Linear and its dx = dy W are independent, and nothing reads dW before the optimizer step (or the
# gradient all-reduce).  The persistent input-gradient GEMM of a 768-wide layer over 50 k tokens has 591 tiles for 256 CUs:
# its third round keeps 79 CUs busy.  Weight-gradient GEMMs that add into gradient sinks are therefore issued on a stream
# of its own: their workgroups (one tile x K slice each, not persistent) start wherever a CU is free and
# fill those rounds, the tails of the attention kernels and the bandwidth-bound kernels in between.  The stream is joined
# at the end of the backward pass (autograd's final callback), before anything can read a gradient.
WGRAD_OVERLAP = os.environ.get("VITK_WGRAD_OVERLAP", "1") != "0"
WGRAD_OVERLAP_MIN_FLOPS = 2e10  # smaller products stay on the main stream (the hand-over costs host time)
_on_wgrad_stream = False  # launches made right now go to the weight-gradient stream (selects its own scratch buffer)
_wgrad_lanes = {}  # device index -> _WgradLane


class _WgradLane:
	"""The weight-gradient stream of one device and what is in flight on it."""

	sections = 0  # launches handed to a second stream so far (all devices)

	def __init__(self, device):
		# a stream of torch's own pool.  (A HIP stream of the lowest priority created outside torch and wrapped as
		# torch.cuda.ExternalStream measured the same step time, but cross-stream waits on it intermittently never returned, or
		# returned early, in a process that had captured HIP graphs before: dropped.)
		self.stream = torch.cuda.Stream(device=device)
		self.held = []  # operands of launches that may still run (the caching allocator must not hand them out again)
		self.hooked = False  # the end-of-backward join is queued
		self.last = None  # event behind the newest launch

	@contextlib.contextmanager
	def section(self, after, *operands):
		"""Launches made inside go to the weight-gradient stream, ordered behind the event `after` (recorded on the main
		stream where the operands were complete) or, without one, behind everything already queued on the current stream.
		Inside a backward pass the join happens at its end; elsewhere, on leaving the section."""
		global _on_wgrad_stream
		if after is not None:
			self.stream.wait_event(after)
		else:
			self.stream.wait_stream(torch.cuda.current_stream())
		if not self.hooked:
			try:
				torch.autograd.Variable._execution_engine.queue_callback(self.join)
				self.hooked = True
			except RuntimeError:  # not inside a backward pass
				pass
		_on_wgrad_stream = True
		_WgradLane.sections += 1
		try:
			with torch.cuda.stream(self.stream):
				yield
		finally:
			_on_wgrad_stream = False
		self.held.append(operands)
		if not self.hooked:
			self.join()

	def join(self):
		"""The current stream waits for everything issued on the weight-gradient stream so far."""
		self.hooked = False
		torch.cuda.current_stream().wait_stream(self.stream)
		self.held.clear()

	def run_behind(self, fn, *args):
		"""fn(*args) with the weight-gradient stream current, ordered behind both streams: a collective that fn launches
		starts after the gradients written on either."""
		self.stream.wait_stream(torch.cuda.current_stream())
		with torch.cuda.stream(self.stream):
			fn(*args)


def _wgrad_lane(t, flops):
	"""The weight-gradient stream for a product of `flops` on t's device, or None if it should run on the current stream."""
	if not WGRAD_OVERLAP or flops < WGRAD_OVERLAP_MIN_FLOPS or GEMM_PROFILE_SERIAL or torch.cuda.is_current_stream_capturing():
		return None
	lane = _wgrad_lanes.get(t.device.index)
	if lane is None:
		lane = _wgrad_lanes[t.device.index] = _WgradLane(t.device)
	return lane


def wgrad_mark(dy, x, dw_out):
	"""An event on the current stream at this point (everything queued so far produces the operands dy, x of a weight-gradient
	product that will be issued later, behind the launch of the input-gradient product that shares its dy), or None when
	that product will stay on the main stream (no gradient sink dw_out, too small, overlap off)."""
	if dw_out is None or _wgrad_lane(dy, 2.0 * dy.shape[0] * dy.shape[1] * x.shape[1]) is None:
		return None
	ev = torch.cuda.Event()
	ev.record()
	return ev


def second_stream(device):
	"""The second stream of `device` (created on first use)."""
	lane = _wgrad_lanes.get(device.index)
	if lane is None:
		lane = _wgrad_lanes[device.index] = _WgradLane(device)
	return lane


def lane_in_flight(device):
	"""The second stream of `device` if launches are in flight on it (weight-gradient products of the backward pass that is
	running), else None."""
	lane = _wgrad_lanes.get(device.index)
	return lane if lane is not None and (lane.held or lane.hooked) else None


def wgrad_sections():
	"""How many weight-gradient products have been handed to a second stream so far (tests, diagnostics)."""
	return _WgradLane.sections


def join_wgrad_stream():
	"""Make the current stream wait for weight-gradient launches still in flight (a no-op when there are none)."""
	for lane in _wgrad_lanes.values():
		if lane.held or lane.hooked:
			lane.join()


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_layout=L.K_MAJOR, b_layout=L.K_MAJOR, bias=None, epilogue=L.EPI_NONE, act=L.ACT_NONE,
		aux=None, ld_aux=0, out2=None, ld_out2=0, gamma=None, accumulate=False, patch=None, tokens_per_image=0, pos=None, a_colsum=None, workspace=True, drop=(), lead_rows=0,
		hold=None):
	"""C[M,N] (+)= epilogue(sum_k A(m,k) B(n,k)); see include/vitk.h for the operand layouts.  drop: up to two (p, seed)
	dropouts fused into the epilogue (entries with p == 0 are ignored).  hold: a list that takes (descriptor, operands)
	instead of a launch (see wgrad_group)."""
	L.require_gpu(A, B, C)
	d = L.GemmDesc()
	d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
	d.M, d.N, d.K = M, N, K
	d.lda, d.ldb, d.ldc = lda, ldb, ldc
	d.a_layout, d.b_layout = a_layout, b_layout
	d.in_dtype, d.out_dtype = L.dtype_code(A.dtype), L.dtype_code(C.dtype)
	assert B.dtype == A.dtype
	d.epilogue, d.act, d.accumulate = epilogue, act, int(accumulate)
	if patch is not None:
		d.patch_grid, d.img_c, d.img_hw = patch
	d.tokens_per_image = tokens_per_image
	d.lead_rows = lead_rows
	d.bias, d.gamma, d.pos = L.ptr(bias), L.ptr(gamma), L.ptr(pos)
	d.aux, d.ld_aux = L.ptr(aux), ld_aux
	d.out2, d.ld_out2 = L.ptr(out2), ld_out2
	d.a_colsum = L.ptr(a_colsum)
	for i, (p, seed) in enumerate([ds for ds in drop if ds[0] > 0.0]):
		d.drop_p[i], d.drop_seed[i] = float(p), int(seed)
	if workspace and a_layout == L.MN_MAJOR and C.dtype == torch.float32:  # weight gradients: the only form that splits K
		ws = gemm_workspace(C.device)
		d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
	assert a_colsum is None or (a_colsum.dtype == torch.float32 and a_colsum.is_contiguous() and a_colsum.numel() == M)
	for t in (bias, gamma, pos):
		assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
	assert aux is None or aux.dtype == A.dtype
	assert out2 is None or out2.dtype == C.dtype
	lib = L.load()
	if hold is not None:
		hold.append((d, (A, B, C, a_colsum)))
		return C
	if GEMM_PROFILE is None:
		L.check(lib.vitk_gemm(d, L.stream()))
		return C
	# kernel timing for the roofline report: HIP events on the launch stream around this one launch
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	L.check(lib.vitk_gemm(d, L.stream()))
	e1.record()
	label = lib.vitk_gemm_last_kernel().decode()  # the kernel family that ran: the name a kernel trace shows
	GEMM_PROFILE.append((label, 2.0 * M * N * K, e0, e1))
	return C


def linear_fwd(x, w, bias=None, act=L.ACT_NONE, residual=None, gamma=None, save_pre=False, out_dtype=None, drop=()):
	"""y = epilogue(x @ w.T + bias).  x: [M, K] (row stride may exceed K), w: [N, K] contiguous.
	Returns y, or (y, pre_activation) with save_pre.  drop: (p, seed) dropouts applied in the epilogue to the activation
	output (act) or to the branch value before the residual add (residual)."""
	M, K = x.shape
	N = w.shape[0]
	assert w.shape[1] == K and x.stride(1) == 1 and w.is_contiguous()
	y = torch.empty((M, N), dtype=out_dtype or x.dtype, device=x.device)
	pre = torch.empty_like(y) if save_pre else None
	if residual is not None:
		assert residual.shape == (M, N) and residual.stride(1) == 1
		gemm(x, w, y, M, N, K, x.stride(0), K, N, bias=bias, epilogue=L.EPI_RESIDUAL, aux=residual, ld_aux=residual.stride(0), gamma=gamma, drop=drop)
	elif act != L.ACT_NONE:
		gemm(x, w, y, M, N, K, x.stride(0), K, N, bias=bias, epilogue=L.EPI_ACT, act=act, out2=pre, ld_out2=N, drop=drop)
	else:
		assert not any(ds[0] > 0.0 for ds in drop), "fused dropout needs an activation or residual epilogue"
		gemm(x, w, y, M, N, K, x.stride(0), K, N, bias=bias)
	return (y, pre) if save_pre else y


def linear_dgrad(dy, w, act=L.ACT_NONE, pre=None, out=None, drop=()):
	"""dx = dy @ w, optionally multiplied by act'(pre) (pre = saved pre-activation of the layer that produced x) and, with
	drop, by the mask of the dropout that followed the activation in the forward pass."""
	M, N = dy.shape
	K = w.shape[1]
	assert w.shape[0] == N and dy.stride(1) == 1 and w.is_contiguous()
	dx = out if out is not None else torch.empty((M, K), dtype=dy.dtype, device=dy.device)
	if DGRAD_TRANSPOSED_WEIGHTS and dy.dtype == torch.bfloat16 and M >= 4096 and N % 64 == 0 and K >= DGRAD_T_MIN and N >= DGRAD_T_MIN:
		# many tokens: a transposed copy of the weight (one small kernel per weight and optimizer step) makes both
		# operands K-major, the layout the large-tile GEMM runs fastest on
		wt = weight_t(w)
		if pre is not None:
			gemm(dy, wt, dx, M, K, N, dy.stride(0), N, dx.stride(0), epilogue=L.EPI_ACTGRAD, act=act, aux=pre, ld_aux=pre.stride(0), drop=drop)
		else:
			gemm(dy, wt, dx, M, K, N, dy.stride(0), N, dx.stride(0))
		return dx
	# sum over n: A = dy (n contiguous), B(n'=k, k'=n) = w[n, k] -> MN-major with ldb = K
	if pre is not None:
		gemm(dy, w, dx, M, K, N, dy.stride(0), K, dx.stride(0), b_layout=L.MN_MAJOR, epilogue=L.EPI_ACTGRAD, act=act, aux=pre, ld_aux=pre.stride(0), drop=drop)
	else:
		gemm(dy, w, dx, M, K, N, dy.stride(0), K, dx.stride(0), b_layout=L.MN_MAJOR)
	return dx


def linear_wgrad(dy, x, bias_grad=False, dw_out=None, db_out=None, overlap=False, after=None, group=True):
	"""dw[N, K] = dy.T @ x in float32.  With bias_grad also db[N] = column sums of dy, formed by the same kernel from
	the dy tiles it stages anyway (or by a column-sum kernel after it, where the GEMM path that runs does not fuse
	it): returns (dw, db).  dw_out / db_out: float32 buffers that the results are ADDED into instead (gradient
	sinks); they are returned in place of fresh tensors.  overlap: the caller writes neither dy nor x afterwards, so a
	large product whose results go to sinks only may run on the weight-gradient stream (joined at the end of backward),
	there ordered behind the event `after` (wgrad_mark) if given, else behind the current stream.  group=False: the caller
	reads the result before its wgrad_group ends, so the product must not be held back."""
	M, N = dy.shape
	K = x.shape[1]
	assert x.shape[0] == M and dy.stride(1) == 1 and x.stride(1) == 1
	dw = dw_out if dw_out is not None else torch.empty((N, K), dtype=torch.float32, device=dy.device)
	db = None
	if bias_grad:
		db = db_out if db_out is not None else torch.zeros(N, dtype=torch.float32, device=dy.device)
	lane = None
	if overlap and dw_out is not None and (db_out is not None or not bias_grad):
		lane = _wgrad_lane(dy, 2.0 * M * N * K)
	hold = None
	if group and lane is None and _wgrad_held is not None and dy.dtype == torch.bfloat16 and 2.0 * M * N * K < WGRAD_GROUP_MAX_FLOPS:
		hold = _wgrad_held
	with (lane.section(after, dy, x) if lane is not None else contextlib.nullcontext()):
		gemm(dy, x, dw, N, K, M, dy.stride(0), x.stride(0), K, a_layout=L.MN_MAJOR, b_layout=L.MN_MAJOR, a_colsum=db, accumulate=dw_out is not None, hold=hold)
	return (dw, db) if bias_grad else dw


# The weight-gradient products of one block of a SMALL model (d_model 192: 4 ... 12 output tiles over tens of thousands of
# tokens each) cannot fill the chip one at a time without cutting K so finely that writing and re-reading the partial
# tiles costs as much as the product.  A block's backward therefore holds them back and hands all of them to the library
# in one call at its end (vitk_gemm_group: one launch, K cut a quarter as finely, one reduction launch).  Large products
# (the ones that go to the second stream) are never held.
WGRAD_GROUP = os.environ.get("VITK_WGRAD_GROUP", "1") != "0"
WGRAD_GROUP_MAX_FLOPS = 2e10  # (the size from which a product alone fills the chip; equals the second stream's lower bound)
GROUP_CALLS = 0  # grouped calls made so far (tests, diagnostics)
_wgrad_held = None  # list of (descriptor, operands) while a block collects
_wgrad_held_done = []  # (parameter, sink) whose "written" reports wait for the held products to be launched


@contextlib.contextmanager
def wgrad_group():
	"""Small weight-gradient products issued inside (linear_wgrad calls that would run on the current stream) are launched
	together on leaving.  Their operands and results must not be touched inside after the call that issued them, and the
	current stream must be the same on leaving."""
	global _wgrad_held, GROUP_CALLS
	if _wgrad_held is not None or not WGRAD_GROUP or GEMM_PROFILE is not None:
		yield
		return
	_wgrad_held = []
	try:
		yield
	except BaseException:
		_wgrad_held = None
		_wgrad_held_done.clear()
		raise
	held, _wgrad_held = _wgrad_held, None
	if len(held) == 1:
		L.check(L.load().vitk_gemm(held[0][0], L.stream()))
	elif held:
		descs = (L.GemmDesc * len(held))(*[h[0] for h in held])
		L.check(L.load().vitk_gemm_group(descs, len(held), L.stream()))
		GROUP_CALLS += 1
	reports = list(_wgrad_held_done)
	_wgrad_held_done.clear()
	for p, sink in reports:
		_sink_done(p, sink)


def colsum(x):
	"""sum over rows -> float32 [N] (bias gradient)."""
	M, N = x.shape
	assert x.stride(1) == 1
	out = torch.empty(N, dtype=torch.float32, device=x.device)
	L.check(L.load().vitk_colsum(L.ptr(x), L.ptr(out), M, N, x.stride(0), 0, L.dtype_code(x.dtype), L.stream()))
	return out


def norm_fwd(x, weight, bias, kind, eps, unit_offset=True, res=None):
	"""kind 'layernorm' | 'rmsnorm'.  x: [rows, d] contiguous.  With res: s = x + res is normalised and returned.
	Returns (y, s_or_None, mean_or_None, rstd)."""
	rows, d = x.shape
	assert x.is_contiguous() and (res is None or res.is_contiguous())
	y = torch.empty_like(x)
	s = torch.empty_like(x) if res is not None else None
	rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
	lib = L.load()
	dt = L.dtype_code(x.dtype)
	if kind == "layernorm":
		mean = torch.empty(rows, dtype=torch.float32, device=x.device)
		L.check(lib.vitk_layernorm_fwd(L.ptr(x), L.ptr(res), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(s), L.ptr(mean), L.ptr(rstd), rows, d, eps, dt, L.stream()))
		return y, s, mean, rstd
	L.check(lib.vitk_rmsnorm_fwd(L.ptr(x), L.ptr(res), L.ptr(weight), L.ptr(y), L.ptr(s), L.ptr(rstd), rows, d, eps, int(unit_offset), dt, L.stream()))
	return y, s, None, rstd


def norm_bwd(dy, x, weight, mean, rstd, kind, unit_offset=True, dres=None, dw_out=None, db_out=None, workspace=True):
	"""Returns (dx, dweight_f32, dbias_f32_or_None); dres (same shape as x) is added into dx.  dw_out / db_out: float32
	[d] buffers the kernel accumulates into instead of fresh zeroed ones (gradient sinks).  workspace=False: the column
	sums are added with float atomics instead of passing through the scratch buffer."""
	rows, d = x.shape
	assert dy.is_contiguous() and x.is_contiguous() and (dres is None or dres.is_contiguous())
	dx = torch.empty_like(x)
	need_b = kind == "layernorm"
	if dw_out is not None and (db_out is not None or not need_b):
		dw, dbuf = dw_out, db_out
	else:
		dwb = torch.zeros((2, d), dtype=torch.float32, device=x.device)  # one fill for both accumulators
		dw = dw_out if dw_out is not None else dwb[0]
		dbuf = db_out if db_out is not None else dwb[1]
	lib = L.load()
	dt = L.dtype_code(x.dtype)
	# the per-workgroup column sums pass through the scratch buffer (stream-ordered with the GEMMs that use it)
	ws_ptr, ws_bytes = 0, 0
	if workspace:
		ws = gemm_workspace(x.device)
		ws_ptr, ws_bytes = ws.data_ptr(), ws.numel()
	if kind == "layernorm":
		db = dbuf
		L.check(lib.vitk_layernorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(weight), L.ptr(mean), L.ptr(rstd), L.ptr(dres), L.ptr(dx), L.ptr(dw), L.ptr(db), rows, d, dt,
			ws_ptr, ws_bytes, L.stream()))
		return dx, dw, db
	L.check(lib.vitk_rmsnorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(weight), L.ptr(rstd), L.ptr(dres), L.ptr(dx), L.ptr(dw), rows, d, int(unit_offset), dt,
		ws_ptr, ws_bytes, L.stream()))
	return dx, dw, None


def attention_fwd(q, k, v, B, n_head, Sq, Skv, dh, scale=0.0, drop=(0.0, 0)):
	"""q: [B*Sq, >=h*dh] view, k/v: [B*Skv, ...] views (unit inner stride).  Returns (out [B*Sq, h*dh], lse [B,h,Sq]).
	drop = (dropout_p, seed): dropout on the attention probabilities (layers/attention.py:41,86); the backward must be
	given the same pair."""
	out = torch.empty((B * Sq, n_head * dh), dtype=q.dtype, device=q.device)
	lse = torch.empty((B, n_head, Sq), dtype=torch.float32, device=q.device)
	assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
	L.check(L.load().vitk_attention_fwd_dropout(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), B, n_head, Sq, Skv, dh,
		q.stride(0), k.stride(0), v.stride(0), out.stride(0), scale, float(drop[0]), int(drop[1]), L.dtype_code(q.dtype), L.stream()))
	return out, lse


def attention_bwd(dout, q, k, v, out, lse, dq, dk, dv, B, n_head, Sq, Skv, dh, scale=0.0, drop=(0.0, 0)):
	"""Writes dq/dk/dv (views with unit inner stride, e.g. column slices of one [rows, 3d] buffer)."""
	assert dout.stride(1) == 1 and dout.stride(0) == out.stride(0)
	scratch = torch.empty((2,) + tuple(lse.shape), dtype=torch.float32, device=lse.device)  # dQ pass -> dK/dV pass
	L.check(L.load().vitk_attention_bwd_dropout(L.ptr(dout), L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(out), L.ptr(lse), L.ptr(dq), L.ptr(dk), L.ptr(dv),
		B, n_head, Sq, Skv, dh, q.stride(0), k.stride(0), v.stride(0), out.stride(0), dq.stride(0), dk.stride(0), dv.stride(0),
		scale, float(drop[0]), int(drop[1]), L.ptr(scratch), L.dtype_code(q.dtype), L.stream()))


def gated_act_fwd(u, F, act):
	M = u.shape[0]
	a = torch.empty((M, F), dtype=u.dtype, device=u.device)
	L.check(L.load().vitk_gated_act_fwd(L.ptr(u), L.ptr(a), M, F, u.stride(0), act, L.dtype_code(u.dtype), L.stream()))
	return a


def gated_act_bwd(da, u, F, act):
	M = u.shape[0]
	du = torch.empty_like(u)
	L.check(L.load().vitk_gated_act_bwd(L.ptr(da), L.ptr(u), L.ptr(du), M, F, u.stride(0), act, L.dtype_code(u.dtype), L.stream()))
	return du


def scale_residual(x, res, gamma):
	y = torch.empty_like(x)
	rows, d = x.shape
	L.check(L.load().vitk_scale_residual(L.ptr(x), L.ptr(res), L.ptr(gamma), L.ptr(y), rows, d, L.dtype_code(x.dtype), L.stream()))
	return y


def dropout_add(x, res, p1, seed1, p2=0.0, seed2=0, out=None):
	"""res + dropout(dropout(x, p1), p2) in one pass (res may be None; out may alias x)."""
	y = torch.empty_like(x) if out is None else out
	L.check(L.load().vitk_dropout_add(L.ptr(x), L.ptr(res), L.ptr(y), x.numel(), float(p1), int(seed1), float(p2), int(seed2), L.dtype_code(x.dtype), L.stream()))
	return y


def layerscale_bwd(dy, branch, gamma):
	rows, d = dy.shape
	dbranch = torch.empty_like(dy)
	dgamma = torch.zeros(d, dtype=torch.float32, device=dy.device)
	L.check(L.load().vitk_layerscale_bwd(L.ptr(dy), L.ptr(branch), L.ptr(gamma), L.ptr(dbranch), L.ptr(dgamma), rows, d, L.dtype_code(dy.dtype), L.stream()))
	return dbranch, dgamma


# --------------------------------------------------------------------------------------------------
# autograd Functions
# --------------------------------------------------------------------------------------------------

def _flat(x):
	x2 = x.reshape(-1, x.shape[-1])
	return x2 if x2.is_contiguous() else x2.contiguous()


class LinearFn(torch.autograd.Function):
	"""y = act(x @ w.T + b) — nn.Linear (+ elementwise activation) on the MFMA GEMM."""

	@staticmethod
	def forward(ctx, x, w, b, act):
		L.require_gpu(x, w)
		dt = compute_dtype(x)
		x2 = _flat(cast(x, dt))
		wc = weight_as(w, dt)
		bc = f32(b)
		if act != L.ACT_NONE:
			y, pre = linear_fwd(x2, wc, bc, act=act, save_pre=True)
		else:
			y, pre = linear_fwd(x2, wc, bc), None
		ctx.save_for_backward(x2, w, pre)
		ctx.meta = (x.shape, x.dtype, b, act, dt)
		return y.reshape(*x.shape[:-1], w.shape[0])

	@staticmethod
	def backward(ctx, dy):
		x2, w, pre = ctx.saved_tensors
		xshape, xdtype, b, act, dt = ctx.meta
		dy2 = _flat(cast(dy, dt))
		if pre is not None:
			du = torch.empty_like(dy2)
			L.check(L.load().vitk_act_bwd(L.ptr(dy2), L.ptr(pre), L.ptr(du), dy2.numel(), act, L.dtype_code(dt), L.stream()))
			dy2 = du
		wc = weight_as(w, dt)
		dx = linear_dgrad(dy2, wc).reshape(xshape) if ctx.needs_input_grad[0] else None
		want_w, want_b = ctx.needs_input_grad[1], b is not None and ctx.needs_input_grad[2]
		sk = _Sinks(w=w if want_w else None, b=b if want_b else None)
		dw = db = None
		if want_w and want_b:  # the bias gradient rides in the weight-gradient kernel
			dw, db = linear_wgrad(dy2, x2, bias_grad=True, dw_out=sk.buf("w"), db_out=sk.buf("b"))
		elif want_w:
			dw = linear_wgrad(dy2, x2, dw_out=sk.buf("w"))
		elif want_b:
			db = colsum(dy2)
			if sk.buf("b") is not None:
				sk.buf("b").add_(db)
		if dx is not None and dx.dtype != xdtype:
			dx = cast(dx, xdtype)
		return dx, sk.ret("w", dw), sk.ret("b", db), None


class NormFn(torch.autograd.Function):
	"""LayerNorm / RMSNorm over the last dimension."""

	@staticmethod
	def forward(ctx, x, weight, bias, kind, eps, unit_offset):
		L.require_gpu(x, weight)
		dt = compute_dtype(x) if kind == "layernorm" else (x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32)
		x2 = _flat(cast(x, dt))
		wf, bf = f32(weight), f32(bias)
		y, _, mean, rstd = norm_fwd(x2, wf, bf, kind, eps, unit_offset)
		ctx.save_for_backward(x2, weight, mean, rstd)
		ctx.meta = (x.shape, x.dtype, bias, kind, unit_offset)
		return y.reshape(x.shape)

	@staticmethod
	def backward(ctx, dy):
		x2, weight, mean, rstd = ctx.saved_tensors
		xshape, xdtype, bias, kind, unit_offset = ctx.meta
		dy2 = _flat(cast(dy, x2.dtype))
		sk = _Sinks(w=weight, b=bias)
		dx, dw, db = norm_bwd(dy2, x2, f32(weight), mean, rstd, kind, unit_offset, dw_out=sk.buf("w"), db_out=sk.buf("b"))
		dx = dx.reshape(xshape)
		if dx.dtype != xdtype:
			dx = cast(dx, xdtype)
		return dx, sk.ret("w", dw), sk.ret("b", db), None, None, None


class DyTFn(torch.autograd.Function):
	"""gamma * tanh(alpha * x) + beta over the last dimension (layers/norm.py:35-45), result in the dtype of x."""

	@staticmethod
	def forward(ctx, x, alpha, gamma, beta):
		L.require_gpu(x, gamma)
		dt = x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32
		x2 = _flat(cast(x, dt))
		y = torch.empty_like(x2)
		L.check(L.load().vitk_dyt_fwd(L.ptr(x2), L.ptr(f32(alpha)), L.ptr(f32(gamma)), L.ptr(f32(beta)), L.ptr(y), x2.shape[0], x2.shape[1],
			L.dtype_code(dt), L.stream()))
		ctx.save_for_backward(x2, alpha, gamma)
		ctx.meta = (x.shape, x.dtype)
		return y.reshape(x.shape)

	@staticmethod
	def backward(ctx, dy):
		x2, alpha, gamma = ctx.saved_tensors
		xshape, xdtype = ctx.meta
		dy2 = _flat(cast(dy, x2.dtype))
		dx = torch.empty_like(x2)
		d = x2.shape[1]
		acc = torch.zeros(2 * d + 1, dtype=torch.float32, device=x2.device)  # [dgamma | dbeta | dalpha]
		L.check(L.load().vitk_dyt_bwd(L.ptr(dy2), L.ptr(x2), L.ptr(f32(alpha)), L.ptr(f32(gamma)), L.ptr(dx), L.ptr(acc[2 * d:]), L.ptr(acc), L.ptr(acc[d:]),
			x2.shape[0], d, L.dtype_code(x2.dtype), L.stream()))
		dx = dx.reshape(xshape)
		if dx.dtype != xdtype:
			dx = cast(dx, xdtype)
		return dx, acc[2 * d:].reshape(alpha.shape).to(alpha.dtype), acc[:d].to(gamma.dtype), acc[d:2 * d].to(gamma.dtype)


def _ffn_forward(x2, w1, b1, w2, b2, act, gated, residual=None, gamma=None, drop1=(0.0, 0), drop_out=()):
	"""fc_in -> act / gate*act -> dropout1 -> fc_out (layers/transformers.py:45-57).  Returns (y, u, a) with `a` the
	input of fc_out (after dropout1: in the activation epilogue of fc_in, or in place after the gating kernel).
	drop_out: dropouts on fc_out's result before the residual add, fused into its epilogue (needs `residual`)."""
	if gated:
		u = linear_fwd(x2, w1, b1)
		F = w2.shape[1]
		a = gated_act_fwd(u, F, act)
		if drop1[0] > 0.0:
			dropout_add(a, None, drop1[0], drop1[1], out=a)
	else:
		a, u = linear_fwd(x2, w1, b1, act=act, save_pre=True, drop=(drop1,))
	y = linear_fwd(a, w2, b2, residual=residual, gamma=gamma, drop=drop_out)
	return y, u, a


def _ffn_backward(dy2, x2, u, a, w1, w2, act, gated, need_dx=True, outs=(None, None, None, None), drop1=(0.0, 0)):
	"""Returns (dx, dw1, db1, dw2, db2) with float32 parameter gradients.  outs = buffers to accumulate
	(dw1, db1, dw2, db2) into (gradient sinks) or None each.  `a` is fc_out's input (after dropout1)."""
	if gated:
		da = linear_dgrad(dy2, w2)
		if drop1[0] > 0.0:
			dropout_add(da, None, drop1[0], drop1[1], out=da)
		du = gated_act_bwd(da, u, w2.shape[1], act)
	else:
		# the dropout mask and the activation derivative are both elementwise factors of the same epilogue
		du = linear_dgrad(dy2, w2, act=act, pre=u, drop=(drop1,))
	# Both weight-gradient products of the branch are issued BEHIND its second input-gradient launch (the persistent kernel
	# gets the CUs first, the weight gradients fill what its partly filled last round leaves idle) and depend on ONE mark
	# (du complete, and dy2 with it): a mark is a marker packet on the main stream and costs the launch behind it 6-7 us
	# (two marks per branch, each weight gradient right behind its own input-gradient launch: 0.14 ms more per ViT-B/16 step)
	ready = wgrad_mark(du, x2, outs[0])
	dx = linear_dgrad(du, w1) if need_dx else None
	dw2, db2 = linear_wgrad(dy2, a, bias_grad=True, dw_out=outs[2], db_out=outs[3], overlap=True, after=ready)
	dw1, db1 = linear_wgrad(du, x2, bias_grad=True, dw_out=outs[0], db_out=outs[1], overlap=True, after=ready)
	return dx, dw1, db1, dw2, db2


class FeedForwardFn(torch.autograd.Function):
	@staticmethod
	def forward(ctx, x, w1, b1, w2, b2, act, gated):
		L.require_gpu(x, w1)
		dt = compute_dtype(x)
		x2 = _flat(cast(x, dt))
		y, u, a = _ffn_forward(x2, weight_as(w1, dt), f32(b1), weight_as(w2, dt), f32(b2), act, gated)
		ctx.save_for_backward(x2, u, a, w1, w2)
		ctx.meta = (x.shape, x.dtype, b1, b2, act, gated, dt)
		return y.reshape(x.shape[:-1] + (w2.shape[0],))

	@staticmethod
	def backward(ctx, dy):
		x2, u, a, w1, w2 = ctx.saved_tensors
		xshape, xdtype, b1, b2, act, gated, dt = ctx.meta
		dy2 = _flat(cast(dy, dt))
		sk = _Sinks(w1=w1, b1=b1, w2=w2, b2=b2)
		with wgrad_group():
			dx, dw1, db1, dw2, db2 = _ffn_backward(dy2, x2, u, a, weight_as(w1, dt), weight_as(w2, dt), act, gated, ctx.needs_input_grad[0],
				outs=(sk.buf("w1"), sk.buf("b1"), sk.buf("w2"), sk.buf("b2")))
		if dx is not None:
			dx = dx.reshape(xshape)
			if dx.dtype != xdtype:
				dx = cast(dx, xdtype)
		return dx, sk.ret("w1", dw1), sk.ret("b1", db1), sk.ret("w2", dw2), sk.ret("b2", db2), None, None


def _mha_forward(x2, wqkv, wo, bo, B, S, n_head, residual=None, gamma=None, drop=(0.0, 0), drop_out=()):
	"""qkv projection -> attention -> out projection (layers/attention.py:27-48).  Returns (y, qkv, o, lse).
	drop: dropout on the attention probabilities; drop_out: dropouts on the projection's result before the residual
	add, fused into its epilogue (needs `residual`)."""
	d = x2.shape[1]
	dh = d // n_head
	qkv = linear_fwd(x2, wqkv)
	o, lse = attention_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, n_head, S, S, dh, drop=drop)
	y = linear_fwd(o, wo, bo, residual=residual, gamma=gamma, drop=drop_out)
	return y, qkv, o, lse


def _mha_backward(dy2, x2, qkv, o, lse, wqkv, wo, B, S, n_head, need_dx=True, outs=(None, None, None), drop=(0.0, 0)):
	"""Returns (dx, dwqkv, dwo, dbo).  outs = buffers to accumulate (dwqkv, dwo, dbo) into (gradient sinks) or None."""
	d = x2.shape[1]
	dh = d // n_head
	do = linear_dgrad(dy2, wo)
	dqkv = torch.empty_like(qkv)
	attention_bwd(do, qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, n_head, S, S, dh, drop=drop)
	ready = wgrad_mark(dqkv, x2, outs[0])  # (one mark for both weight-gradient products, behind the input-gradient launch: see _ffn_backward)
	dx = linear_dgrad(dqkv, wqkv) if need_dx else None
	dwo, dbo = linear_wgrad(dy2, o, bias_grad=True, dw_out=outs[1], db_out=outs[2], overlap=True, after=ready)
	dwqkv = linear_wgrad(dqkv, x2, dw_out=outs[0], overlap=True, after=ready)
	return dx, dwqkv, dwo, dbo


class MHAFn(torch.autograd.Function):
	@staticmethod
	def forward(ctx, x, wqkv, wo, bo, n_head, drop=(0.0, 0)):
		L.require_gpu(x, wqkv)
		dt = compute_dtype(x)
		B, S, d = x.shape
		x2 = _flat(cast(x, dt))
		y, qkv, o, lse = _mha_forward(x2, weight_as(wqkv, dt), weight_as(wo, dt), f32(bo), B, S, n_head, drop=drop)
		ctx.save_for_backward(x2, qkv, o, lse, wqkv, wo)
		ctx.meta = (x.shape, x.dtype, bo, n_head, dt, drop)
		return y.reshape(B, S, d)

	@staticmethod
	def backward(ctx, dy):
		x2, qkv, o, lse, wqkv, wo = ctx.saved_tensors
		(B, S, d), xdtype, bo, n_head, dt, drop = ctx.meta
		dy2 = _flat(cast(dy, dt))
		sk = _Sinks(wqkv=wqkv, wo=wo, bo=bo)
		with wgrad_group():
			dx, dwqkv, dwo, dbo = _mha_backward(dy2, x2, qkv, o, lse, weight_as(wqkv, dt), weight_as(wo, dt), B, S, n_head, ctx.needs_input_grad[0],
				outs=(sk.buf("wqkv"), sk.buf("wo"), sk.buf("bo")), drop=drop)
		if dx is not None:
			dx = dx.reshape(B, S, d)
			if dx.dtype != xdtype:
				dx = cast(dx, xdtype)
		return dx, sk.ret("wqkv", dwqkv), sk.ret("wo", dwo), sk.ret("bo", dbo), None, None


class ViTBlockFn(torch.autograd.Function):
	"""One pre-norm transformer block (layers/vit.py:27-57, MHA attention) as a single autograd node:
	norm -> qkv GEMM -> attention -> out-proj GEMM (+bias, +residual in the epilogue) -> norm ->
	fc_in GEMM (+bias, +activation in the epilogue) -> fc_out GEMM (+bias, +residual in the epilogue).
	With LayerScale the residual add runs as a separate kernel because the branch output is needed
	for the LayerScale gradient."""

	@staticmethod
	def forward(ctx, x, n1w, n1b, wqkv, wo, bo, ls1, n2w, n2b, w1, b1, w2, b2, ls2, cfg):
		L.require_gpu(x, wqkv)
		kind, eps, unit_offset, n_head, act, gated = cfg[:6]
		# active dropouts of the block (None = none): {"attn": attention probabilities, "a": attention branch,
		# "f1" / "f2": FeedForward.dropout1 / dropout2, "b": feed-forward branch}, each (p, seed); without LayerScale only
		dr = cfg[6] if len(cfg) > 6 and cfg[6] is not None else {}
		off = (0.0, 0)
		d_attn, d_a, d_f1, d_f2, d_b = (dr.get(k, off) for k in ("attn", "a", "f1", "f2", "b"))
		assert not dr or (ls1 is None and ls2 is None)
		dt = compute_dtype(x)
		B, S, d = x.shape
		x0 = _flat(cast(x, dt))
		wqkv_c, wo_c, w1_c, w2_c = (weight_as(t, dt) for t in (wqkv, wo, w1, w2))
		n1, _, mean1, rstd1 = norm_fwd(x0, f32(n1w), f32(n1b), kind, eps, unit_offset)
		if ls1 is None:  # residual add (and the branch dropout, if any) in the projection's epilogue
			x1, qkv, o, lse = _mha_forward(n1, wqkv_c, wo_c, f32(bo), B, S, n_head, residual=x0, drop=d_attn, drop_out=(d_a,))
			br1 = None
		else:
			br1, qkv, o, lse = _mha_forward(n1, wqkv_c, wo_c, f32(bo), B, S, n_head)
			x1 = scale_residual(br1, x0, f32(ls1))
		n2, _, mean2, rstd2 = norm_fwd(x1, f32(n2w), f32(n2b), kind, eps, unit_offset)
		if ls2 is None:
			x2, u, a = _ffn_forward(n2, w1_c, f32(b1), w2_c, f32(b2), act, gated, residual=x1, drop1=d_f1, drop_out=(d_f2, d_b))
			br2 = None
		else:
			br2, u, a = _ffn_forward(n2, w1_c, f32(b1), w2_c, f32(b2), act, gated)
			x2 = scale_residual(br2, x1, f32(ls2))
		ctx.save_for_backward(x0, n1, mean1, rstd1, qkv, o, lse, br1, x1, n2, mean2, rstd2, u, a, br2,
			n1w, wqkv, wo, ls1, n2w, w1, w2, ls2)
		ctx.meta = (x.shape, x.dtype, n1b, bo, n2b, b1, b2, cfg, dt)
		return x2.reshape(B, S, d)

	@staticmethod
	def backward(ctx, dy):
		(x0, n1, mean1, rstd1, qkv, o, lse, br1, x1, n2, mean2, rstd2, u, a, br2, n1w, wqkv, wo, ls1, n2w, w1, w2, ls2) = ctx.saved_tensors
		(B, S, d), xdtype, n1b, bo, n2b, b1, b2, cfg, dt = ctx.meta
		kind, eps, unit_offset, n_head, act, gated = cfg[:6]
		dr = cfg[6] if len(cfg) > 6 and cfg[6] is not None else {}
		off = (0.0, 0)
		d_attn, d_a, d_f1, d_f2, d_b = (dr.get(k, off) for k in ("attn", "a", "f1", "f2", "b"))
		wqkv_c, wo_c, w1_c, w2_c = (weight_as(t, dt) for t in (wqkv, wo, w1, w2))
		dx2 = _flat(cast(dy, dt))
		# parameters with a registered gradient sink get their gradients accumulated in place by the kernels
		sk = {n: _sink(t) for n, t in (("n1w", n1w), ("n1b", n1b), ("wqkv", wqkv), ("wo", wo), ("bo", bo), ("n2w", n2w), ("n2b", n2b), ("w1", w1), ("b1", b1),
			("w2", w2), ("b2", b2))}
		buf = {n: (v[0] if v is not None else None) for n, v in sk.items()}
		with wgrad_group():  # (a small model's four weight-gradient products leave together at the end)
			# feed-forward branch
			if ls2 is not None:
				dbr2, dls2 = layerscale_bwd(dx2, br2, f32(ls2))
			elif d_f2[0] > 0.0 or d_b[0] > 0.0:
				m1, m2 = (d_f2, d_b) if d_f2[0] > 0.0 else (d_b, off)
				dbr2, dls2 = dropout_add(dx2, None, m1[0], m1[1], m2[0], m2[1]), None
			else:
				dbr2, dls2 = dx2, None
			dn2, dw1, db1, dw2, db2 = _ffn_backward(dbr2, n2, u, a, w1_c, w2_c, act, gated, outs=(buf["w1"], buf["b1"], buf["w2"], buf["b2"]), drop1=d_f1)
			dx1, dn2w, dn2b = norm_bwd(dn2, x1, f32(n2w), mean2, rstd2, kind, unit_offset, dres=dx2, dw_out=buf["n2w"], db_out=buf["n2b"])
			# attention branch
			if ls1 is not None:
				dbr1, dls1 = layerscale_bwd(dx1, br1, f32(ls1))
			elif d_a[0] > 0.0:
				dbr1, dls1 = dropout_add(dx1, None, d_a[0], d_a[1]), None
			else:
				dbr1, dls1 = dx1, None
			dn1, dwqkv, dwo, dbo = _mha_backward(dbr1, n1, qkv, o, lse, wqkv_c, wo_c, B, S, n_head, outs=(buf["wqkv"], buf["wo"], buf["bo"]), drop=d_attn)
			dx0, dn1w, dn1b = norm_bwd(dn1, x0, f32(n1w), mean1, rstd1, kind, unit_offset, dres=dx1, dw_out=buf["n1w"], db_out=buf["n1b"])
		dx0 = dx0.reshape(B, S, d)
		if dx0.dtype != xdtype:
			dx0 = cast(dx0, xdtype)

		def out(name, g, p):
			return _sink_done(p, sk[name]) if sk[name] is not None else grad_as(g, p)
		return (dx0, out("n1w", dn1w, n1w), out("n1b", dn1b, n1b), out("wqkv", dwqkv, wqkv), out("wo", dwo, wo), out("bo", dbo, bo), grad_as(dls1, ls1),
			out("n2w", dn2w, n2w), out("n2b", dn2b, n2b), out("w1", dw1, w1), out("b1", db1, b1), out("w2", dw2, w2), out("b2", db2, b2), grad_as(dls2, ls2), None)


def add_rows(dst, src):
	"""dst[r, :] += src[r, :] for two [rows, d] views with unit inner stride (any row strides)."""
	rows, d = src.shape
	assert dst.shape == src.shape and dst.stride(1) == 1 and src.stride(1) == 1 and dst.dtype == src.dtype
	L.check(L.load().vitk_add_rows(L.ptr(dst), L.ptr(src), rows, d, dst.stride(0), src.stride(0), L.dtype_code(dst.dtype), L.stream()))


def _dgrad_rows(dy, w, r0, r1):
	"""dy @ w[r0:r1, :] for a Linear weight w [out, in] in the compute dtype (dy: [M, r1 - r0]): the input gradient through
	the output features r0 .. r1 of a fused projection.  Many bf16 tokens read the kept transposed copy, as linear_dgrad."""
	M, N = dy.shape
	K = w.shape[1]
	assert N == r1 - r0 and dy.stride(1) == 1
	if DGRAD_TRANSPOSED_WEIGHTS and dy.dtype == torch.bfloat16 and M >= 4096 and N % 64 == 0 and K >= DGRAD_T_MIN and N >= DGRAD_T_MIN:
		wt = weight_t(w)  # [in, out]: B(n = input feature, k = output feature) K-major with row stride `out`
		dx = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
		gemm(dy, wt[:, r0:r1], dx, M, K, N, dy.stride(0), wt.stride(0), K)
		return dx
	return linear_dgrad(dy, w[r0:r1])


class ViTLastBlockFn(torch.autograd.Function):
	"""The LAST pre-norm block of a ViT whose head reads the class-token row only (models/vit.py:54-65 there: the loop over
	the blocks, then `out = out[:, 0]`).  Of the last block's output only row 0 of every sample is ever read, so of its
	work only what that row depends on is done: the norm and the key / value projection of ALL tokens (the class token's
	query attends to every key), and for the class rows alone the query projection, the attention row, the output
	projection, the second norm and the feed-forward.  Logits, loss and every parameter / input gradient are those of
	the full block (the rows left out have zero upstream gradient in the reference as well); 10 of the block's 12
	d x d-sized products shrink from B * S rows to B.  Returns [B, 1, d].  No LayerScale, no active dropout (the caller
	falls back to ViTBlockFn)."""

	@staticmethod
	def forward(ctx, x, n1w, n1b, wqkv, wo, bo, n2w, n2b, w1, b1, w2, b2, cfg):
		L.require_gpu(x, wqkv)
		kind, eps, unit_offset, n_head, act, gated = cfg[:6]
		dt = compute_dtype(x)
		B, S, d = x.shape
		dh = d // n_head
		x0 = _flat(cast(x, dt))
		wqkv_c, wo_c, w1_c, w2_c = (weight_as(t, dt) for t in (wqkv, wo, w1, w2))
		n1, _, mean1, rstd1 = norm_fwd(x0, f32(n1w), f32(n1b), kind, eps, unit_offset)
		kv = linear_fwd(n1, wqkv_c[d:])  # keys and values of every token
		q = linear_fwd(n1[::S], wqkv_c[:d])  # queries of the class rows (read in place through the row stride S * d)
		o, lse = attention_fwd(q, kv[:, :d], kv[:, d:], B, n_head, 1, S, dh)
		x1 = linear_fwd(o, wo_c, f32(bo), residual=x0[::S])
		n2, _, mean2, rstd2 = norm_fwd(x1, f32(n2w), f32(n2b), kind, eps, unit_offset)
		x2, u, a = _ffn_forward(n2, w1_c, f32(b1), w2_c, f32(b2), act, gated, residual=x1)
		ctx.save_for_backward(x0, n1, mean1, rstd1, kv, q, o, lse, x1, n2, mean2, rstd2, u, a, n1w, wqkv, wo, n2w, w1, w2)
		ctx.meta = (x.shape, x.dtype, n1b, bo, n2b, b1, b2, cfg, dt)
		return x2.reshape(B, 1, d)

	@staticmethod
	def backward(ctx, dy):
		(x0, n1, mean1, rstd1, kv, q, o, lse, x1, n2, mean2, rstd2, u, a, n1w, wqkv, wo, n2w, w1, w2) = ctx.saved_tensors
		(B, S, d), xdtype, n1b, bo, n2b, b1, b2, cfg, dt = ctx.meta
		kind, eps, unit_offset, n_head, act, gated = cfg[:6]
		dh = d // n_head
		wqkv_c, wo_c, w1_c, w2_c = (weight_as(t, dt) for t in (wqkv, wo, w1, w2))
		dx2 = _flat(cast(dy, dt))  # [B, d]
		sk = {n: _sink(t) for n, t in (("n1w", n1w), ("n1b", n1b), ("wqkv", wqkv), ("wo", wo), ("bo", bo), ("n2w", n2w), ("n2b", n2b), ("w1", w1), ("b1", b1),
			("w2", w2), ("b2", b2))}
		buf = {n: (v[0] if v is not None else None) for n, v in sk.items()}
		# class rows: feed-forward branch, second norm, output projection, the attention row
		dn2, dw1, db1, dw2, db2 = _ffn_backward(dx2, n2, u, a, w1_c, w2_c, act, gated, outs=(buf["w1"], buf["b1"], buf["w2"], buf["b2"]))
		dx1, dn2w, dn2b = norm_bwd(dn2, x1, f32(n2w), mean2, rstd2, kind, unit_offset, dres=dx2, dw_out=buf["n2w"], db_out=buf["n2b"])
		dwo, dbo = linear_wgrad(dx1, o, bias_grad=True, dw_out=buf["wo"], db_out=buf["bo"])
		do = linear_dgrad(dx1, wo_c)
		dq = torch.empty_like(q)
		dkv = torch.empty_like(kv)
		attention_bwd(do, q, kv[:, :d], kv[:, d:], o, lse, dq, dkv[:, :d], dkv[:, d:], B, n_head, 1, S, dh)
		# fused qkv projection: weight rows [0, d) from the class rows' queries, rows [d, 3d) from every token's keys and values
		dwqkv = buf["wqkv"] if buf["wqkv"] is not None else torch.zeros((3 * d, d), dtype=torch.float32, device=dy.device)
		ready = wgrad_mark(dkv, n1, buf["wqkv"])
		dn1 = _dgrad_rows(dkv, wqkv_c, d, 3 * d)
		linear_wgrad(dkv, n1, dw_out=dwqkv[d:], overlap=buf["wqkv"] is not None, after=ready)
		linear_wgrad(dq, n1[::S], dw_out=dwqkv[:d])
		add_rows(dn1[::S], _dgrad_rows(dq, wqkv_c, 0, d))
		dx0, dn1w, dn1b = norm_bwd(dn1, x0, f32(n1w), mean1, rstd1, kind, unit_offset, dw_out=buf["n1w"], db_out=buf["n1b"])
		add_rows(dx0[::S], dx1)  # the residual path exists for the class rows only
		dx0 = dx0.reshape(B, S, d)
		if dx0.dtype != xdtype:
			dx0 = cast(dx0, xdtype)

		def out(name, g, p):
			return _sink_done(p, sk[name]) if sk[name] is not None else grad_as(g, p)
		return (dx0, out("n1w", dn1w, n1w), out("n1b", dn1b, n1b), out("wqkv", dwqkv, wqkv), out("wo", dwo, wo), out("bo", dbo, bo),
			out("n2w", dn2w, n2w), out("n2b", dn2b, n2b), out("w1", dw1, w1), out("b1", db1, b1), out("w2", dw2, w2), out("b2", db2, b2), None)


class PatchEmbedFn(torch.autograd.Function):
	"""images -> tokens [B, L + P*P, d]: im2col-free patch GEMM + bias + pos_embed in the epilogue, the L leading learned
	token rows written by a small kernel (models/vit.py:46-52: L = 1, the class token; with `dist_token` L = 2, the
	class token and a distillation token, the student of a DeiT-style distillation)."""

	@staticmethod
	def forward(ctx, images, w, b, class_token, pos_embed, n_patch, dist_token=None):
		L.require_gpu(images, w)
		dt = compute_dtype(images)
		Bn, C, H, W = images.shape
		assert H == W, "Patchify method is implemented for square images only"
		img = cast(images, dt)
		P2 = n_patch * n_patch
		lead = 1 if dist_token is None else 2
		d = w.shape[0]
		K = w.shape[1]
		x = torch.empty((Bn, P2 + lead, d), dtype=dt, device=images.device)
		pos = f32(pos_embed).reshape(P2 + lead, d)
		lib, dtc = L.load(), L.dtype_code(dt)
		patches = None
		if dt == torch.bfloat16 and (H // n_patch) % 8 != 0:
			# the bf16 MFMA kernels gather patches by 16-byte pieces of a pixel row; narrower patches (4 x 4 on 32 x 32 inputs)
			# are laid out as rows first (a few MB, kept for the weight gradient) and take the plain MFMA GEMM
			patches = torch.empty((Bn * P2, K), dtype=dt, device=images.device)
			L.check(lib.vitk_patchify(L.ptr(img), L.ptr(patches), Bn, C, H, n_patch, dtc, L.stream()))
			gemm(patches, weight_as(w, dt), x, Bn * P2, d, K, K, K, d, bias=f32(b), tokens_per_image=P2, pos=pos, lead_rows=lead)
		else:
			gemm(img, weight_as(w, dt), x, Bn * P2, d, K, K, K, d, bias=f32(b), patch=(n_patch, C, H), tokens_per_image=P2, pos=pos, lead_rows=lead)
		L.check(lib.vitk_cls_rows(L.ptr(x), L.ptr(f32(class_token)), L.ptr(pos), Bn, P2 + lead, d, dtc, L.stream()))
		if dist_token is not None:  # row 1 of every image: the same kernel on the buffers shifted by one row
			L.check(lib.vitk_cls_rows(x.data_ptr() + d * x.element_size(), L.ptr(f32(dist_token)), pos.data_ptr() + 4 * d, Bn, P2 + lead, d, dtc, L.stream()))
		ctx.save_for_backward(img, w)
		ctx.patches = patches
		ctx.meta = (images.shape, images.dtype, b, class_token, pos_embed, n_patch, dt, dist_token)
		return x

	@staticmethod
	def backward(ctx, dx):
		img, w = ctx.saved_tensors
		(Bn, C, H, W), idtype, b, class_token, pos_embed, n_patch, dt, dist_token = ctx.meta
		lib = L.load()
		P2 = n_patch * n_patch
		lead = 1 if dist_token is None else 2
		d, K = w.shape
		dx = cast(dx, dt)
		dtc = L.dtype_code(dt)
		sk = _Sinks(w=w, b=b, cls=class_token, pos=pos_embed, dist=dist_token)
		S = P2 + lead
		# d pos_embed = sum over the batch of dx; d class_token (d dist_token) = the same sum of token row 0 (1) alone, read
		# through the row stride.  Each goes straight into its gradient sink where there is one.
		dpos = sk.buf("pos") if sk.buf("pos") is not None else torch.empty((S, d), dtype=torch.float32, device=dx.device)
		L.check(lib.vitk_batch_sum(L.ptr(dx), L.ptr(dpos), Bn, S, d, int(sk.buf("pos") is not None), dtc, L.stream()))

		def lead_row_grad(name, param, r):
			buf = sk.buf(name)
			out = buf if buf is not None else torch.empty(d, dtype=torch.float32, device=dx.device)
			L.check(lib.vitk_colsum(dx.data_ptr() + r * d * dx.element_size(), L.ptr(out), Bn, d, S * d, int(buf is not None), dtc, L.stream()))
			return sk.ret(name, out.reshape(param.shape))
		dcls = lead_row_grad("cls", class_token, 0)
		ddist = None if dist_token is None else lead_row_grad("dist", dist_token, 1)
		dtok = torch.empty((Bn * P2, d), dtype=dt, device=dx.device)
		L.check(lib.vitk_drop_lead_rows(L.ptr(dx), L.ptr(dtok), Bn, S, d, lead, dtc, L.stream()))
		patches = ctx.patches
		if patches is None:
			patches = torch.empty((Bn * P2, K), dtype=dt, device=dx.device)
			L.check(lib.vitk_patchify(L.ptr(img), L.ptr(patches), Bn, C, H, n_patch, dtc, L.stream()))
		dw, db = linear_wgrad(dtok, patches, bias_grad=True, dw_out=sk.buf("w"), db_out=sk.buf("b"))
		dimg = None
		if ctx.needs_input_grad[0]:
			dpatch = linear_dgrad(dtok, weight_as(w, dt))
			dimg = torch.empty((Bn, C, H, W), dtype=dt, device=dx.device)
			L.check(lib.vitk_unpatchify(L.ptr(dpatch), L.ptr(dimg), Bn, C, H, n_patch, dtc, L.stream()))
			if dimg.dtype != idtype:
				dimg = cast(dimg, idtype)
		return (dimg, sk.ret("w", dw), sk.ret("b", db), dcls, sk.ret("pos", dpos.reshape(pos_embed.shape)), None, ddist)


class DualHeadFn(torch.autograd.Function):
	"""(x[:, 0] @ w_cls.T + b_cls, x[:, 1] @ w_dist.T + b_dist): classifier on the class-token row (models/vit.py:63-65) and a
	second head on the distillation-token row, both by ONE bandwidth-bound launch that reads the two rows in place."""

	@staticmethod
	def forward(ctx, x, w_cls, b_cls, w_dist, b_dist):
		L.require_gpu(x, w_cls, w_dist)
		dt = compute_dtype(x)
		xc = cast(x, dt)
		Bn, S, d = xc.shape
		C = w_cls.shape[0]
		assert S >= 2 and w_dist.shape == w_cls.shape == (C, d)
		y0 = torch.empty((Bn, C), dtype=dt, device=x.device)
		y1 = torch.empty((Bn, C), dtype=dt, device=x.device)
		L.check(L.load().vitk_dual_head_fwd(L.ptr(xc), L.ptr(weight_as(w_cls, dt)), L.ptr(f32(b_cls)), L.ptr(weight_as(w_dist, dt)), L.ptr(f32(b_dist)), L.ptr(y0),
			L.ptr(y1), Bn, S, d, C, L.dtype_code(dt), L.stream()))
		ctx.save_for_backward(xc, w_cls, w_dist)
		ctx.meta = (x.dtype, b_cls, b_dist, dt)
		return y0, y1

	@staticmethod
	def backward(ctx, dy0, dy1):
		xc, w_cls, w_dist = ctx.saved_tensors
		xdtype, b_cls, b_dist, dt = ctx.meta
		Bn, S, d = xc.shape
		rows = xc.reshape(Bn * S, d)
		dx = torch.zeros((Bn, S, d), dtype=dt, device=xc.device) if ctx.needs_input_grad[0] else None
		grads = []
		for r, dy, w, b in ((0, dy0, w_cls, b_cls), (1, dy1, w_dist, b_dist)):
			dyc = cast(dy, dt)
			sk = _Sinks(w=w, b=b)
			dw, db = linear_wgrad(dyc, rows[r::S], bias_grad=True, dw_out=sk.buf("w"), db_out=sk.buf("b"))
			if dx is not None:
				linear_dgrad(dyc, weight_as(w, dt), out=dx.reshape(Bn * S, d)[r::S])
			grads += [sk.ret("w", dw), sk.ret("b", db)]
		if dx is not None and dx.dtype != xdtype:
			dx = cast(dx, xdtype)
		return (dx, *grads)


class ClsHeadFn(torch.autograd.Function):
	"""logits = x[:, 0] @ w.T + b (models/vit.py:63-65): the class-token row is read in place through the row stride."""

	@staticmethod
	def forward(ctx, x, w, b):
		L.require_gpu(x, w)
		dt = compute_dtype(x)
		xc = cast(x, dt)
		Bn, S, d = xc.shape
		rows0 = xc.reshape(Bn * S, d)[::S]  # [B, d] view with row stride S*d
		y = linear_fwd(rows0, weight_as(w, dt), f32(b))
		ctx.save_for_backward(xc, w)
		ctx.meta = (x.dtype, b, dt)
		return y

	@staticmethod
	def backward(ctx, dy):
		xc, w = ctx.saved_tensors
		xdtype, b, dt = ctx.meta
		Bn, S, d = xc.shape
		dy2 = cast(dy, dt)
		rows0 = xc.reshape(Bn * S, d)[::S]
		sk = _Sinks(w=w, b=b)
		dw, db = linear_wgrad(dy2, rows0, bias_grad=True, dw_out=sk.buf("w"), db_out=sk.buf("b"))
		dx = None
		if ctx.needs_input_grad[0]:
			dx = torch.zeros((Bn, S, d), dtype=dt, device=dy.device)
			linear_dgrad(dy2, weight_as(w, dt), out=dx.reshape(Bn * S, d)[::S])
			if dx.dtype != xdtype:
				dx = cast(dx, xdtype)
		return dx, sk.ret("w", dw), sk.ret("b", db)


class CeKdLossFn(torch.autograd.Function):
	"""(1-alpha) * CE(logits, y; label_smoothing) + alpha * tau^2 * KL(softmax(t/tau) || softmax(s/tau)), batchmean
	(main.py:106-109, deit_trainer.py:30-45).  teacher=None gives plain cross entropy."""

	@staticmethod
	def forward(ctx, logits, labels, teacher, label_smoothing, alpha, tau):
		L.require_gpu(logits, labels)
		lg = logits.contiguous()
		if lg.dtype not in (torch.float32, torch.bfloat16):
			raise TypeError(lg.dtype)
		t = None if teacher is None else cast(teacher.detach(), lg.dtype)
		Bn, C = lg.shape
		loss = torch.zeros(1, dtype=torch.float32, device=lg.device)
		dlogits = torch.empty_like(lg)
		L.check(L.load().vitk_ce_kd_loss(L.ptr(lg), L.ptr(labels.contiguous()), L.ptr(t), L.ptr(loss), L.ptr(dlogits), Bn, C,
			float(label_smoothing), float(alpha), float(tau), L.dtype_code(lg.dtype), L.stream()))
		ctx.save_for_backward(dlogits)
		return loss[0]

	@staticmethod
	def backward(ctx, g):
		(dlogits,) = ctx.saved_tensors
		return scale_by(dlogits, g), None, None, None, None, None


class DeitDualLossFn(torch.autograd.Function):
	"""(1-alpha) * CE(logits_cls, y; label_smoothing) + alpha * tau^2 * KL(softmax(teacher/tau) || softmax(logits_dist/tau)),
	batchmean: the loss of deit_trainer.py:30-45 with the soft term on the distillation head."""

	@staticmethod
	def forward(ctx, logits_cls, logits_dist, labels, teacher, label_smoothing, alpha, tau):
		L.require_gpu(logits_cls, logits_dist, labels, teacher)
		lc, ld = logits_cls.contiguous(), logits_dist.contiguous()
		assert lc.dtype == ld.dtype and lc.shape == ld.shape
		t = cast(teacher.detach(), lc.dtype)
		Bn, C = lc.shape
		loss = torch.zeros(1, dtype=torch.float32, device=lc.device)
		dc, dd = torch.empty_like(lc), torch.empty_like(ld)
		L.check(L.load().vitk_deit_dual_loss(L.ptr(lc), L.ptr(ld), L.ptr(labels.contiguous()), L.ptr(t), L.ptr(loss), L.ptr(dc), L.ptr(dd), Bn, C,
			float(label_smoothing), float(alpha), float(tau), L.dtype_code(lc.dtype), L.stream()))
		ctx.save_for_backward(dc, dd)
		return loss[0]

	@staticmethod
	def backward(ctx, g):
		dc, dd = ctx.saved_tensors
		return scale_by(dc, g), scale_by(dd, g), None, None, None, None, None


_epoch_owner = {}  # device index -> the DropoutEpoch registered with the library on that device


class DropoutEpoch:
	"""A device word that every dropout mask drawn on this device mixes into its seed while it is registered
	(vitk_dropout_epoch_set).  A training step captured into a HIP graph keeps the seeds it was captured with; it calls
	advance() once at its start instead, so every replay draws new masks and the forward and backward of one replay the same
	ones.  While the word holds 0 the masks are those of the seeds alone.  One word per device: a later instance replaces
	an earlier one for new launches (launches already captured keep reading the word they were captured with, which
	therefore lives as long as its owner).  The registered instance is kept alive by this module until close()."""

	def __init__(self, device):
		self.device = torch.device(device)
		self.word = torch.zeros(1, dtype=torch.int32, device=self.device)
		with torch.cuda.device(self.device):
			L.check(L.load().vitk_dropout_epoch_set(self.word.data_ptr()))
		_epoch_owner[self.device.index] = self

	def advance(self):
		L.check(L.load().vitk_dropout_epoch_advance(self.word.data_ptr(), L.stream()))

	def set(self, value):
		self.word.fill_(int(value))

	def close(self):
		if _epoch_owner.get(self.device.index) is self:
			del _epoch_owner[self.device.index]
			with torch.cuda.device(self.device):
				L.check(L.load().vitk_dropout_epoch_set(None))


def scale_by(t, g):
	"""t * g for the scalar upstream gradient g (a 0-dim device tensor) of a loss."""
	out = torch.empty_like(t)
	L.check(L.load().vitk_scale_by(L.ptr(t), L.ptr(cast(g.reshape(1), torch.float32)), L.ptr(out), t.numel(), L.dtype_code(t.dtype), L.stream()))
	return out


def deit_dual_loss(logits_cls, logits_dist, labels, teacher, alpha=0.5, tau=1.0, label_smoothing=0.0):
	return DeitDualLossFn.apply(logits_cls, logits_dist, labels, teacher, label_smoothing, alpha, tau)


def cross_entropy(logits, labels, label_smoothing=0.0):
	return CeKdLossFn.apply(logits, labels, None, label_smoothing, 0.0, 1.0)


def deit_loss(student, labels, teacher, alpha=0.5, tau=1.0, label_smoothing=0.0):
	return CeKdLossFn.apply(student, labels, teacher, label_smoothing, alpha, tau)


class GatedActFn(torch.autograd.Function):
	"""a = gate * act(up) for u = [gate | up] (layers/transformers.py:48-50)."""

	@staticmethod
	def forward(ctx, u, act):
		u2 = _flat(u)
		F = u2.shape[1] // 2
		ctx.save_for_backward(u2)
		ctx.meta = (u.shape, act, F)
		return gated_act_fwd(u2, F, act).reshape(u.shape[:-1] + (F,))

	@staticmethod
	def backward(ctx, da):
		(u2,) = ctx.saved_tensors
		ushape, act, F = ctx.meta
		return gated_act_bwd(_flat(cast(da, u2.dtype)), u2, F, act).reshape(ushape), None


class ScaleResidualFn(torch.autograd.Function):
	"""x + gamma * branch (layers/vit.py:44-47,52-55); gamma None = plain residual add."""

	@staticmethod
	def forward(ctx, branch, x, gamma):
		L.require_gpu(branch, x)
		dt = compute_dtype(x) if x.dtype != branch.dtype else x.dtype
		b2, x2 = _flat(cast(branch, dt)), _flat(cast(x, dt))
		ctx.save_for_backward(b2 if gamma is not None else None, gamma)
		ctx.meta = (x.shape, x.dtype, branch.dtype, dt)
		return scale_residual(b2, x2, f32(gamma)).reshape(x.shape)

	@staticmethod
	def backward(ctx, dy):
		b2, gamma = ctx.saved_tensors
		xshape, xdtype, bdtype, dt = ctx.meta
		dy2 = _flat(cast(dy, dt))
		if gamma is None:
			dbranch, dgamma = dy2, None
		else:
			dbranch, dgamma = layerscale_bwd(dy2, b2, f32(gamma))
		dbranch = dbranch.reshape(xshape)
		dx = dy2.reshape(xshape)
		return (dbranch if dbranch.dtype == bdtype else cast(dbranch, bdtype)), (dx if dx.dtype == xdtype else cast(dx, xdtype)), grad_as(dgamma, gamma)


class PatchifyFn(torch.autograd.Function):
	"""Non-overlapping patch gather 'b c (p1 hp) (p2 wp) -> b (p1 p2) (c hp wp)' (layers/embeddings.py:7-16)."""

	@staticmethod
	def forward(ctx, images, n_patch):
		L.require_gpu(images)
		Bn, C, H, W = images.shape
		img = images.contiguous()
		p = H // n_patch
		out = torch.empty((Bn, n_patch * n_patch, C * p * p), dtype=img.dtype, device=img.device)
		L.check(L.load().vitk_patchify(L.ptr(img), L.ptr(out), Bn, C, H, n_patch, L.dtype_code(img.dtype), L.stream()))
		ctx.meta = (images.shape, n_patch)
		return out

	@staticmethod
	def backward(ctx, dp):
		(Bn, C, H, W), n_patch = ctx.meta
		dp = dp.contiguous()
		dimg = torch.empty((Bn, C, H, W), dtype=dp.dtype, device=dp.device)
		L.check(L.load().vitk_unpatchify(L.ptr(dp), L.ptr(dimg), Bn, C, H, n_patch, L.dtype_code(dp.dtype), L.stream()))
		return dimg, None


_dropout_calls = [0]


class DropoutFn(torch.autograd.Function):
	@staticmethod
	def forward(ctx, x, p, seed):
		L.require_gpu(x)
		xc = x.contiguous()
		y = torch.empty_like(xc)
		L.check(L.load().vitk_dropout(L.ptr(xc), L.ptr(y), xc.numel(), p, seed, L.dtype_code(xc.dtype), L.stream()))
		ctx.meta = (p, seed)
		return y

	@staticmethod
	def backward(ctx, dy):
		p, seed = ctx.meta
		dyc = dy.contiguous()
		dx = torch.empty_like(dyc)
		L.check(L.load().vitk_dropout(L.ptr(dyc), L.ptr(dx), dyc.numel(), p, seed, L.dtype_code(dyc.dtype), L.stream()))
		return dx, None, None


def next_dropout_seed():
	"""Seed of the next dropout mask: torch's CPU generator seed (so torch.manual_seed makes runs repeatable) mixed with
	a per-call counter; no device synchronisation."""
	_dropout_calls[0] += 1
	return (torch.initial_seed() * 0x9E3779B1 + _dropout_calls[0] * 0x85EBCA77) & 0xFFFFFFFFFFFFFFFF


def seed_counter():
	return _dropout_calls[0]


def set_seed_counter(n):
	_dropout_calls[0] = int(n)


def dropout(x, p):
	"""Training-mode inverted dropout."""
	return DropoutFn.apply(x, float(p), next_dropout_seed())


# --------------------------------------------------------------------------------------------------
# CvT: conv token embedding, conv-projection attention, mean pool
# --------------------------------------------------------------------------------------------------

BN_ACT_FUSED = os.environ.get("VITK_BN_ACT_FUSED", "1") != "0"  # A-B switch: BatchNorm + activation of an MBConv in one pass


def batchnorm_act_fusable(x):
	"""Whether batchnorm_fwd / batchnorm_bwd take `act` for x [rows, C] (16-byte channel groups)."""
	return BN_ACT_FUSED and x.shape[1] % (8 if x.dtype == torch.bfloat16 else 4) == 0 and x.is_contiguous()


def batchnorm_fwd(x, gamma, beta, running_mean, running_var, y, save_mean, save_rstd, eps, momentum, training, act=L.ACT_NONE):
	"""vitk_batchnorm_fwd_ws over x [rows, C]: the column sums pass through the GEMM scratch buffer and are added in a fixed
	order (bit-reproducible batch statistics).  act: y = act(batchnorm(x)) in the same pass (batchnorm_act_fusable)."""
	rows, C = x.shape
	ws = gemm_workspace(x.device)
	assert ws.numel() >= L.load().vitk_batchnorm_workspace_bytes(rows, C), "scratch buffer smaller than the BatchNorm partial sums"
	if act != L.ACT_NONE:
		L.check(L.load().vitk_batchnorm_act_fwd(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var), L.ptr(y), L.ptr(save_mean),
			L.ptr(save_rstd), rows, C, eps, momentum, int(training), act, L.dtype_code(x.dtype), ws.data_ptr(), ws.numel(), L.stream()))
		return
	L.check(L.load().vitk_batchnorm_fwd_ws(L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(running_mean), L.ptr(running_var), L.ptr(y), L.ptr(save_mean), L.ptr(save_rstd),
		rows, C, eps, momentum, int(training), L.dtype_code(x.dtype), ws.data_ptr(), ws.numel(), L.stream()))


def batchnorm_bwd(dy, x, gamma, save_mean, save_rstd, dx, dgamma, dbeta, training, act=L.ACT_NONE, beta=None):
	"""act (with beta): dy is the gradient behind the activation that batchnorm_fwd fused."""
	rows, C = x.shape
	ws = gemm_workspace(x.device)
	assert ws.numel() >= L.load().vitk_batchnorm_workspace_bytes(rows, C), "scratch buffer smaller than the BatchNorm partial sums"
	if act != L.ACT_NONE:
		L.check(L.load().vitk_batchnorm_act_bwd(L.ptr(dy), L.ptr(x), L.ptr(gamma), L.ptr(beta), L.ptr(save_mean), L.ptr(save_rstd), L.ptr(dx), L.ptr(dgamma),
			L.ptr(dbeta), rows, C, int(training), act, L.dtype_code(x.dtype), ws.data_ptr(), ws.numel(), L.stream()))
		return
	L.check(L.load().vitk_batchnorm_bwd_ws(L.ptr(dy), L.ptr(x), L.ptr(gamma), L.ptr(save_mean), L.ptr(save_rstd), L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta),
		rows, C, int(training), L.dtype_code(x.dtype), ws.data_ptr(), ws.numel(), L.stream()))


def _conv_out(n, ks, stride, pad):
	return (n + 2 * pad - ks) // stride + 1


class ConvEmbedFn(torch.autograd.Function):
	"""Conv2d(bias=False) -> BatchNorm2d -> tokens (layers/embeddings.py:24-28 of the reference) as
	im2col gather + MFMA GEMM + token-major BatchNorm.  Input: NCHW, or token-major [B, H*W, C] with hw=(H, W)."""

	@staticmethod
	def forward(ctx, x, conv_w, bn_w, bn_b, running_mean, running_var, cfg):
		ks, stride, pad, eps, momentum, training, hw = cfg
		L.require_gpu(x, conv_w)
		lib = L.load()
		dt = compute_dtype(x)
		xc = cast(x, dt)
		if hw is None:
			Bn, C, H, W = xc.shape
			tokens_in = 0
		else:
			Bn, S, C = xc.shape
			H, W = hw
			assert S == H * W
			tokens_in = 1
		Ho, Wo = _conv_out(H, ks, stride, pad), _conv_out(W, ks, stride, pad)
		Cout = conv_w.shape[0]
		K = C * ks * ks
		rows = Bn * Ho * Wo
		dtc = L.dtype_code(dt)
		# K = 3 * 7 * 7 = 147 for the first CvT stage: zero columns up to a multiple of 8 (written by the gather itself, and a
		# zero-padded copy of the weight) put all three GEMMs of this layer on the bf16 MFMA kernels (the weight gradient is a
		# [64, 147] x 16384-deep product: 3 workgroups and no K split on the generic kernel)
		Kp = (K + 7) // 8 * 8 if dt == torch.bfloat16 else K
		cols = torch.empty((rows, Kp), dtype=dt, device=x.device)
		L.check(lib.vitk_im2col(L.ptr(xc), L.ptr(cols), Bn, C, H, W, ks, stride, pad, Kp, tokens_in, dtc, L.stream()))
		w2 = weight_padded(conv_w, dt, Cout, K, Kp)
		y0 = linear_fwd(cols, w2)
		y = torch.empty_like(y0)
		save_mean = torch.empty(Cout, dtype=torch.float32, device=x.device)
		save_rstd = torch.empty(Cout, dtype=torch.float32, device=x.device)
		batchnorm_fwd(y0, f32(bn_w), f32(bn_b), running_mean, running_var, y, save_mean, save_rstd, eps, momentum, training)
		ctx.save_for_backward(cols, y0, save_mean, save_rstd, conv_w, bn_w)
		ctx.meta = (x.shape, x.dtype, bn_b, cfg, dt, (Bn, C, H, W, Ho, Wo, tokens_in))
		return y.reshape(Bn, Ho * Wo, Cout)

	@staticmethod
	def backward(ctx, dy):
		cols, y0, save_mean, save_rstd, conv_w, bn_w = ctx.saved_tensors
		xshape, xdtype, bn_b, cfg, dt, (Bn, C, H, W, Ho, Wo, tokens_in) = ctx.meta
		ks, stride, pad, eps, momentum, training, hw = cfg
		lib = L.load()
		dtc = L.dtype_code(dt)
		rows, Cout = y0.shape
		dy2 = _flat(cast(dy, dt))
		dy0 = torch.empty_like(y0)
		dgamma = torch.empty(Cout, dtype=torch.float32, device=dy.device)
		dbeta = torch.empty(Cout, dtype=torch.float32, device=dy.device)
		batchnorm_bwd(dy2, y0, f32(bn_w), save_mean, save_rstd, dy0, dgamma, dbeta, training)
		K = conv_w[0].numel()
		Kp = cols.shape[1]  # K, or K padded with zero columns
		dw = linear_wgrad(dy0, cols)
		dw = (dw[:, :K] if Kp != K else dw).reshape(conv_w.shape)
		dx = None
		if ctx.needs_input_grad[0]:
			dcols = linear_dgrad(dy0, weight_padded(conv_w, dt, Cout, K, Kp))
			dx = torch.empty(xshape, dtype=dt, device=dy.device)
			L.check(lib.vitk_col2im(L.ptr(dcols), L.ptr(dx), Bn, C, H, W, ks, stride, pad, Kp, tokens_in, dtc, L.stream()))
			if dx.dtype != xdtype:
				dx = cast(dx, xdtype)
		return dx, grad_as(dw, conv_w), grad_as(dgamma, bn_w), grad_as(dbeta, bn_b), None, None, None


def conv_embedding(mod, x, hw=None):
	"""Forward of a layers.embeddings.ConvEmbedding module (parameters in mod.conv / mod.norm)."""
	conv, bn = mod.conv, mod.norm
	ks, stride, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
	assert conv.kernel_size[0] == conv.kernel_size[1] and conv.bias is None and conv.groups == 1
	training = bn.training or not bn.track_running_stats
	if bn.momentum is None:
		raise NotImplementedError("BatchNorm2d(momentum=None) (cumulative average) is not supported")
	if training and bn.track_running_stats:
		bn.num_batches_tracked += 1
	return ConvEmbedFn.apply(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, (ks, stride, pad, bn.eps, bn.momentum, training, hw))


def dwconv_fwd(x, w, Bn, H, W, C, ks, stride, pad):
	"""Depthwise convolution of token-major x [Bn * H * W, C]; x may be a column slice of a wider buffer (row stride > C)."""
	Ho, Wo = _conv_out(H, ks, stride, pad), _conv_out(W, ks, stride, pad)
	assert x.stride(1) == 1 and x.shape == (Bn * H * W, C)
	y = torch.empty((Bn * Ho * Wo, C), dtype=x.dtype, device=x.device)
	L.check(L.load().vitk_dwconv_fwd_ld(L.ptr(x), x.stride(0), L.ptr(w), L.ptr(y), Bn, H, W, C, ks, stride, pad, L.dtype_code(x.dtype), L.stream()))
	return y, Ho * Wo


def dwconv_bwd(dy, x, w, Bn, H, W, C, ks, stride, pad, dx=None, dw_out=None):
	"""(dx, dw) of dwconv_fwd; x and the optional output dx [Bn * H * W, C] may be column slices of wider buffers.  dw_out:
	a float32 [C, 1, ks, ks] buffer the weight gradient is ADDED into (a gradient sink), returned in place of a fresh tensor."""
	assert x.stride(1) == 1 and dy.is_contiguous()
	if dx is None:
		dx = torch.empty((Bn * H * W, C), dtype=x.dtype, device=x.device)
	assert dx.stride(1) == 1 and dx.shape == (Bn * H * W, C)
	if dw_out is not None:
		assert dw_out.dtype == torch.float32 and dw_out.is_contiguous() and dw_out.numel() == C * ks * ks
	dw = dw_out if dw_out is not None else torch.empty((C, 1, ks, ks), dtype=torch.float32, device=x.device)
	L.check(L.load().vitk_dwconv_bwd_acc(L.ptr(dy), L.ptr(x), x.stride(0), L.ptr(w), L.ptr(dx), dx.stride(0), L.ptr(dw), int(dw_out is not None), Bn, H, W, C, ks,
		stride, pad, L.dtype_code(x.dtype), L.stream()))
	return dx, dw


DWCONV_SINK = os.environ.get("VITK_DWCONV_SINK", "1") != "0"  # A-B switch: depthwise-convolution weight gradients straight into their sinks


def _joint(ts):
	"""The tensors stacked along their first dimension as ONE tensor without a copy, if they are equal-shaped, contiguous and
	lie back to back inside one storage (parameters of a flat buffer laid out by dp.FlatModel, their bfloat16 copies and
	their gradient sinks do); else None."""
	if any(t is None for t in ts):
		return None
	t0 = ts[0]
	step = t0.numel() * t0.element_size()
	for i, t in enumerate(ts):
		if t.shape != t0.shape or t.dtype != t0.dtype or not t.is_contiguous() or t.data_ptr() != t0.data_ptr() + i * step:
			return None
	if t0.untyped_storage().nbytes() < t0.storage_offset() * t0.element_size() + len(ts) * step:
		return None  # neighbours by accident: separate allocations
	return t0.as_strided((len(ts) * t0.shape[0],) + tuple(t0.shape[1:]), t0.stride())


def _stacked(ts):
	"""_joint(ts), or a stacked copy (one kernel) where the tensors are apart."""
	j = _joint(ts)
	return j if j is not None else torch.cat(ts, dim=0)


def _conv_attn_forward(x2, wqkv, bqkv, ckv, wo, bo, geom, residual=None, drop=(0.0, 0)):
	"""Joint q | k | v projection -> depthwise convolution of K | V -> attention -> output projection (+ residual in its
	epilogue).  geom = (Bn, H, W, n_head, ks, stride, pad).  Returns (y, qkv, kv1, o, lse, Skv)."""
	Bn, H, W, n_head, ks, stride, pad = geom
	C = x2.shape[1]
	qkv = linear_fwd(x2, wqkv, bqkv)  # [tokens, 3C]: q | k | v
	kv1, Skv = dwconv_fwd(qkv[:, C:], ckv, Bn, H, W, 2 * C, ks, stride, pad)  # [Bn * Skv, 2C]: k | v
	o, lse = attention_fwd(qkv[:, :C], kv1[:, :C], kv1[:, C:], Bn, n_head, H * W, Skv, C // n_head, drop=drop)
	y = linear_fwd(o, wo, bo, residual=residual)
	return y, qkv, kv1, o, lse, Skv


def _conv_attn_backward(dy2, x2, qkv, kv1, o, lse, wqkv, ckv, wo, geom, Skv, sk, need_dx=True, drop=(0.0, 0)):
	"""Backward of _conv_attn_forward.  sk: a _Sinks over the names wq bq wk bk wv bv ck cv wo bo.  Returns (dx, [dwq, dwk, dwv],
	[dbq, dbk, dbv], [dck, dcv], dwo, dbo) with the parameter gradients as autograd wants them (None where a sink took them)."""
	Bn, H, W, n_head, ks, stride, pad = geom
	C = x2.shape[1]
	dwo, dbo = linear_wgrad(dy2, o, bias_grad=True, dw_out=sk.buf("wo"), db_out=sk.buf("bo"), overlap=True)
	do = linear_dgrad(dy2, wo)
	dqkv = torch.empty_like(qkv)
	dkv1 = torch.empty_like(kv1)
	attention_bwd(do, qkv[:, :C], kv1[:, :C], kv1[:, C:], o, lse, dqkv[:, :C], dkv1[:, :C], dkv1[:, C:], Bn, n_head, H * W, Skv, C // n_head, drop=drop)
	c_sink = _joint([sk.buf("ck"), sk.buf("cv")]) if DWCONV_SINK else None  # (neighbours in a flat buffer: the kernel adds into both at once)
	_, dckv = dwconv_bwd(dkv1, qkv[:, C:], ckv, Bn, H, W, 2 * C, ks, stride, pad, dx=dqkv[:, C:], dw_out=c_sink)
	# the stacked weight / bias gradients go straight into the three neighbouring sinks where there are such; slices of
	# a gradient whose sinks are apart (or absent) are added by hand (or handed to autograd)
	w_sink = _joint([sk.buf("wq"), sk.buf("wk"), sk.buf("wv")])
	b_sink = _joint([sk.buf("bq"), sk.buf("bk"), sk.buf("bv")])
	by_hand = (w_sink is None and any(sk.buf(n) is not None for n in ("wq", "wk", "wv"))) or (b_sink is None and any(sk.buf(n) is not None for n in ("bq", "bk", "bv")))
	dw3, db3 = linear_wgrad(dqkv, x2, bias_grad=True, dw_out=w_sink, db_out=b_sink, group=not by_hand, overlap=not by_hand)  # (dqkv is complete: the convolution's input gradient went into it above)

	def hand_out(names, joint_sink, g3):
		out = []
		for i, name in enumerate(names):
			if joint_sink is None and sk.buf(name) is not None:
				sk.buf(name).add_(g3[i * C:(i + 1) * C])
			out.append(sk.ret(name, g3[i * C:(i + 1) * C]))
		return out
	gw = hand_out(("wq", "wk", "wv"), w_sink, dw3)
	gb = hand_out(("bq", "bk", "bv"), b_sink, db3)
	gc = hand_out(("ck", "cv"), c_sink, dckv)
	dx = None
	if need_dx:
		# dx = [dq | dk | dv] @ [Wq; Wk; Wv]: the weight as an MN-major operand (row n of the stacked weight is k-index n)
		dx = torch.empty((x2.shape[0], C), dtype=x2.dtype, device=x2.device)
		gemm(dqkv, wqkv, dx, x2.shape[0], C, 3 * C, 3 * C, C, C, b_layout=L.MN_MAJOR)
	return dx, gw, gb, gc, sk.ret("wo", dwo), sk.ret("bo", dbo)


class ConvAttentionFn(torch.autograd.Function):
	"""CvT attention (layers/attention.py:67-93 of the reference): q/k/v Linear (with bias), depthwise conv on K and V
	applied directly to token-major activations, attention with Skv <= Sq, output projection.

	The three projections read the same input, so they run as ONE GEMM against the stacked weight [3C, C] into a [tokens, 3C]
	buffer whose column slices are q, k, v (the depthwise convolutions and the attention kernels take row strides); K and V
	share ONE depthwise convolution over 2C channels; backward mirrors it: one weight-gradient GEMM, one input-gradient GEMM
	with K = 3C.  Inside a dp.FlatModel the three weights (and biases, and convolution weights) are neighbours in the flat
	buffers, so the stacked operands, their bf16 copies and their gradient sinks are views; elsewhere they are
	concatenated (one small kernel each) and the gradients are handed to autograd as slices."""

	@staticmethod
	def forward(ctx, x, wq, bq, wk, bk, wv, bv, ck, cv, wo, bo, cfg):
		H, W, n_head, ks, stride, pad = cfg[:6]
		drop = cfg[6] if len(cfg) > 6 else (0.0, 0)
		L.require_gpu(x, wq)
		dt = compute_dtype(x)
		Bn, S, C = x.shape
		assert S == H * W
		x2 = _flat(cast(x, dt))
		wqkv = _stacked([weight_as(wq, dt), weight_as(wk, dt), weight_as(wv, dt)])
		bqkv = _stacked([f32(bq), f32(bk), f32(bv)])
		ckv = _stacked([weight_as(ck, dt), weight_as(cv, dt)])
		geom = (Bn, H, W, n_head, ks, stride, pad)
		y, qkv, kv1, o, lse, Skv = _conv_attn_forward(x2, wqkv, bqkv, ckv, weight_as(wo, dt), f32(bo), geom, drop=drop)
		ctx.save_for_backward(x2, qkv, kv1, o, lse, wq, wk, wv, ck, cv, wo, wqkv, ckv)
		ctx.meta = (x.shape, x.dtype, (bq, bk, bv, bo), geom, dt, Skv, drop)
		return y.reshape(Bn, S, C)

	@staticmethod
	def backward(ctx, dy):
		x2, qkv, kv1, o, lse, wq, wk, wv, ck, cv, wo, wqkv, ckv = ctx.saved_tensors
		(Bn, S, C), xdtype, (bq, bk, bv, bo), geom, dt, Skv, drop = ctx.meta
		dy2 = _flat(cast(dy, dt))
		sk = _Sinks(wq=wq, bq=bq, wk=wk, bk=bk, wv=wv, bv=bv, ck=ck, cv=cv, wo=wo, bo=bo)
		with wgrad_group():
			dx, gw, gb, gc, gwo, gbo = _conv_attn_backward(dy2, x2, qkv, kv1, o, lse, wqkv, ckv, weight_as(wo, dt), geom, Skv, sk, ctx.needs_input_grad[0], drop)
		if dx is not None:
			dx = dx.reshape(Bn, S, C)
			if dx.dtype != xdtype:
				dx = cast(dx, xdtype)
		return (dx, gw[0], gb[0], gw[1], gb[1], gw[2], gb[2], gc[0], gc[1], gwo, gbo, None)


class CvTBlockFn(torch.autograd.Function):
	"""One CvT block (layers/cvt.py:8-50 of the reference) as a single autograd node: norm -> joint q | k | v GEMM ->
	depthwise convolution of K | V -> attention -> output projection (+ bias, + residual in the epilogue) -> norm ->
	fc_in GEMM (+ bias, + activation) -> fc_out GEMM (+ bias, + residual in the epilogue); the backward adds the residual
	gradients inside the norm backward kernels.  With LayerScale the residual add is a separate kernel (the branch output
	is needed for the LayerScale gradient).  No dropout: a block with an active dropout runs module by module."""

	@staticmethod
	def forward(ctx, x, n1w, n1b, wq, bq, wk, bk, wv, bv, ck, cv, wo, bo, ls1, n2w, n2b, w1, b1, w2, b2, ls2, cfg):
		L.require_gpu(x, wq)
		kind, eps, unit_offset, H, W, n_head, ks, stride, pad, act, gated = cfg
		dt = compute_dtype(x)
		Bn, S, C = x.shape
		assert S == H * W
		x0 = _flat(cast(x, dt))
		wqkv = _stacked([weight_as(wq, dt), weight_as(wk, dt), weight_as(wv, dt)])
		bqkv = _stacked([f32(bq), f32(bk), f32(bv)])
		ckv = _stacked([weight_as(ck, dt), weight_as(cv, dt)])
		geom = (Bn, H, W, n_head, ks, stride, pad)
		n1, _, mean1, rstd1 = norm_fwd(x0, f32(n1w), f32(n1b), kind, eps, unit_offset)
		if ls1 is None:
			x1, qkv, kv1, o, lse, Skv = _conv_attn_forward(n1, wqkv, bqkv, ckv, weight_as(wo, dt), f32(bo), geom, residual=x0)
			br1 = None
		else:
			br1, qkv, kv1, o, lse, Skv = _conv_attn_forward(n1, wqkv, bqkv, ckv, weight_as(wo, dt), f32(bo), geom)
			x1 = scale_residual(br1, x0, f32(ls1))
		n2, _, mean2, rstd2 = norm_fwd(x1, f32(n2w), f32(n2b), kind, eps, unit_offset)
		w1_c, w2_c = weight_as(w1, dt), weight_as(w2, dt)
		if ls2 is None:
			x2, u, a = _ffn_forward(n2, w1_c, f32(b1), w2_c, f32(b2), act, gated, residual=x1)
			br2 = None
		else:
			br2, u, a = _ffn_forward(n2, w1_c, f32(b1), w2_c, f32(b2), act, gated)
			x2 = scale_residual(br2, x1, f32(ls2))
		ctx.save_for_backward(x0, n1, mean1, rstd1, qkv, kv1, o, lse, br1, x1, n2, mean2, rstd2, u, a, br2, wqkv, ckv,
			n1w, wq, wk, wv, ck, cv, wo, ls1, n2w, w1, w2, ls2)
		ctx.meta = (x.shape, x.dtype, (n1b, bq, bk, bv, bo, n2b, b1, b2), cfg, dt, geom, Skv)
		return x2.reshape(Bn, S, C)

	@staticmethod
	def backward(ctx, dy):
		(x0, n1, mean1, rstd1, qkv, kv1, o, lse, br1, x1, n2, mean2, rstd2, u, a, br2, wqkv, ckv,
			n1w, wq, wk, wv, ck, cv, wo, ls1, n2w, w1, w2, ls2) = ctx.saved_tensors
		(Bn, S, C), xdtype, (n1b, bq, bk, bv, bo, n2b, b1, b2), cfg, dt, geom, Skv = ctx.meta
		kind, eps, unit_offset, H, W, n_head, ks, stride, pad, act, gated = cfg
		sk = _Sinks(n1w=n1w, n1b=n1b, wq=wq, bq=bq, wk=wk, bk=bk, wv=wv, bv=bv, ck=ck, cv=cv, wo=wo, bo=bo, n2w=n2w, n2b=n2b, w1=w1, b1=b1, w2=w2, b2=b2)
		dx2 = _flat(cast(dy, dt))
		with wgrad_group():  # (the block's four weight-gradient products leave together at the end)
			# feed-forward branch
			dbr2, dls2 = layerscale_bwd(dx2, br2, f32(ls2)) if ls2 is not None else (dx2, None)
			dn2, dw1, db1, dw2, db2 = _ffn_backward(dbr2, n2, u, a, weight_as(w1, dt), weight_as(w2, dt), act, gated,
				outs=(sk.buf("w1"), sk.buf("b1"), sk.buf("w2"), sk.buf("b2")))
			dx1, dn2w, dn2b = norm_bwd(dn2, x1, f32(n2w), mean2, rstd2, kind, unit_offset, dres=dx2, dw_out=sk.buf("n2w"), db_out=sk.buf("n2b"))
			# attention branch
			dbr1, dls1 = layerscale_bwd(dx1, br1, f32(ls1)) if ls1 is not None else (dx1, None)
			dn1, gw, gb, gc, gwo, gbo = _conv_attn_backward(dbr1, n1, qkv, kv1, o, lse, wqkv, ckv, weight_as(wo, dt), geom, Skv, sk)
			dx0, dn1w, dn1b = norm_bwd(dn1, x0, f32(n1w), mean1, rstd1, kind, unit_offset, dres=dx1, dw_out=sk.buf("n1w"), db_out=sk.buf("n1b"))
		dx0 = dx0.reshape(Bn, S, C)
		if dx0.dtype != xdtype:
			dx0 = cast(dx0, xdtype)
		return (dx0, sk.ret("n1w", dn1w), sk.ret("n1b", dn1b), gw[0], gb[0], gw[1], gb[1], gw[2], gb[2], gc[0], gc[1],
			gwo, gbo, grad_as(dls1, ls1), sk.ret("n2w", dn2w), sk.ret("n2b", dn2b), sk.ret("w1", dw1), sk.ret("b1", db1), sk.ret("w2", dw2), sk.ret("b2", db2),
			grad_as(dls2, ls2), None)


class MBConvFn(torch.autograd.Function):
	"""MBConv (layers/conv.py:20-44 of the reference) on token-major activations: Conv2d 1x1 -> BatchNorm2d -> act ->
	depthwise Conv2d 3x3 (stride, pad 1) -> BatchNorm2d -> act -> Conv2d 1x1 -> BatchNorm2d (+ input).  NCHW in and out
	as in the reference; inside, the 1x1 convolutions are MFMA GEMMs over [B*H*W, C] rows, the depthwise convolution
	and the BatchNorms run on the same rows, and the NCHW <-> rows gathers are the k = 1 cases of im2col / col2im."""

	@staticmethod
	def forward(ctx, x, w1, g1, b1, rm1, rv1, wd, g2, b2, rm2, rv2, w3, g3, b3, rm3, rv3, cfg):
		stride, act, use_res, eps, momentum, training = cfg[:6]
		hw = cfg[6] if len(cfg) > 6 else None  # (H, W): x and the result are token-major [B, H*W, C] (no NCHW gathers at all)
		L.require_gpu(x, w1)
		lib = L.load()
		dt = compute_dtype(x)
		dtc = L.dtype_code(dt)
		xc = cast(x, dt).contiguous()
		hid, Cout = w1.shape[0], w3.shape[0]
		if hw is None:
			Bn, Cin, H, W = xc.shape
			R = Bn * H * W
			t0 = torch.empty((R, Cin), dtype=dt, device=x.device)
			L.check(lib.vitk_im2col(L.ptr(xc), L.ptr(t0), Bn, Cin, H, W, 1, 1, 0, Cin, 0, dtc, L.stream()))
		else:
			Bn, S, Cin = xc.shape
			H, W = hw
			assert S == H * W
			R = Bn * S
			t0 = xc.reshape(R, Cin)

		def bn(h, g, b, rm, rv, fused_act=L.ACT_NONE):
			y = torch.empty_like(h)
			mean = torch.empty(h.shape[1], dtype=torch.float32, device=h.device)
			rstd = torch.empty(h.shape[1], dtype=torch.float32, device=h.device)
			batchnorm_fwd(h, f32(g), f32(b), rm, rv, y, mean, rstd, eps, momentum, training, act=fused_act)
			return y, mean, rstd

		def act_fwd(u):
			a = torch.empty_like(u)
			L.check(lib.vitk_act_fwd(L.ptr(u), L.ptr(a), u.numel(), act, dtc, L.stream()))
			return a
		h1 = linear_fwd(t0, weight_as(w1, dt).reshape(hid, Cin))
		# BatchNorm + activation in one pass where the channels allow (n1 / n2, the normalised values, then do not exist: the
		# backward forms them again inside the BatchNorm backward kernels)
		fused = act != L.ACT_NONE and batchnorm_act_fusable(h1)
		if fused:
			a1, m1, r1 = bn(h1, g1, b1, rm1, rv1, act)
			n1 = None
		else:
			n1, m1, r1 = bn(h1, g1, b1, rm1, rv1)
			a1 = act_fwd(n1)
		h2, S2 = dwconv_fwd(a1, weight_as(wd, dt), Bn, H, W, hid, 3, stride, 1)
		if fused:
			a2, m2, r2 = bn(h2, g2, b2, rm2, rv2, act)
			n2 = None
		else:
			n2, m2, r2 = bn(h2, g2, b2, rm2, rv2)
			a2 = act_fwd(n2)
		h3 = linear_fwd(a2, weight_as(w3, dt).reshape(Cout, hid))
		n3, m3, r3 = bn(h3, g3, b3, rm3, rv3)
		if use_res:
			n3 = scale_residual(n3, t0, None)
		Ho, Wo = _conv_out(H, 3, stride, 1), _conv_out(W, 3, stride, 1)
		if hw is None:
			y = torch.empty((Bn, Cout, Ho, Wo), dtype=dt, device=x.device)
			L.check(lib.vitk_col2im(L.ptr(n3), L.ptr(y), Bn, Cout, Ho, Wo, 1, 1, 0, Cout, 0, dtc, L.stream()))
		else:
			y = n3.reshape(Bn, Ho * Wo, Cout)
		ctx.save_for_backward(t0, h1, n1, a1, h2, n2, a2, h3, m1, r1, m2, r2, m3, r3, w1, wd, w3, g1, g2, g3)
		ctx.meta = (x.shape, x.dtype, (b1, b2, b3), cfg, dt, (Bn, Cin, H, W, hid, Cout, Ho, Wo))
		ctx.fused = fused
		return y

	@staticmethod
	def backward(ctx, dy):
		t0, h1, n1, a1, h2, n2, a2, h3, m1, r1, m2, r2, m3, r3, w1, wd, w3, g1, g2, g3 = ctx.saved_tensors
		fused = ctx.fused
		xshape, xdtype, (b1, b2, b3), cfg, dt, (Bn, Cin, H, W, hid, Cout, Ho, Wo) = ctx.meta
		stride, act, use_res, eps, momentum, training = cfg[:6]
		hw = cfg[6] if len(cfg) > 6 else None
		lib = L.load()
		dtc = L.dtype_code(dt)
		dyc = cast(dy, dt).contiguous()
		if hw is None:
			d3 = torch.empty((Bn * Ho * Wo, Cout), dtype=dt, device=dy.device)
			L.check(lib.vitk_im2col(L.ptr(dyc), L.ptr(d3), Bn, Cout, Ho, Wo, 1, 1, 0, Cout, 0, dtc, L.stream()))
		else:
			d3 = dyc.reshape(Bn * Ho * Wo, Cout)

		def bn_bwd(dn, h, g, mean, rstd, fused_act=L.ACT_NONE, b=None):
			dh = torch.empty_like(h)
			dg = torch.empty(h.shape[1], dtype=torch.float32, device=h.device)
			db = torch.empty(h.shape[1], dtype=torch.float32, device=h.device)
			batchnorm_bwd(dn, h, f32(g), mean, rstd, dh, dg, db, training, act=fused_act, beta=f32(b) if b is not None else None)
			return dh, dg, db

		def act_bwd(da, pre):
			du = torch.empty_like(pre)
			L.check(lib.vitk_act_bwd(L.ptr(da), L.ptr(pre), L.ptr(du), pre.numel(), act, dtc, L.stream()))
			return du
		dh3, dg3, db3 = bn_bwd(d3, h3, g3, m3, r3)
		dw3 = linear_wgrad(dh3, a2).reshape(w3.shape)
		da2 = linear_dgrad(dh3, weight_as(w3, dt).reshape(Cout, hid))
		if fused:
			dh2, dg2, db2 = bn_bwd(da2, h2, g2, m2, r2, act, b2)
		else:
			dh2, dg2, db2 = bn_bwd(act_bwd(da2, n2), h2, g2, m2, r2)
		da1, dwd = dwconv_bwd(dh2, a1, weight_as(wd, dt), Bn, H, W, hid, 3, stride, 1)
		if fused:
			dh1, dg1, db1 = bn_bwd(da1, h1, g1, m1, r1, act, b1)
		else:
			dh1, dg1, db1 = bn_bwd(act_bwd(da1, n1), h1, g1, m1, r1)
		dw1 = linear_wgrad(dh1, t0).reshape(w1.shape)
		dx = None
		if ctx.needs_input_grad[0]:
			dt0 = linear_dgrad(dh1, weight_as(w1, dt).reshape(hid, Cin))
			if use_res:
				dt0 = scale_residual(dt0, d3, None)
			if hw is None:
				dx = torch.empty(xshape, dtype=dt, device=dy.device)
				L.check(lib.vitk_col2im(L.ptr(dt0), L.ptr(dx), Bn, Cin, H, W, 1, 1, 0, Cin, 0, dtc, L.stream()))
			else:
				dx = dt0.reshape(xshape)
			if dx.dtype != xdtype:
				dx = cast(dx, xdtype)
		return (dx, grad_as(dw1, w1), grad_as(dg1, g1), grad_as(db1, b1), None, None, grad_as(dwd, wd), grad_as(dg2, g2), grad_as(db2, b2), None, None,
			grad_as(dw3, w3), grad_as(dg3, g3), grad_as(db3, b3), None, None, None)


class ActFn(torch.autograd.Function):
	"""y = act(x) as a stand-alone kernel (the activation behind a BatchNorm, where no GEMM epilogue can take it)."""

	@staticmethod
	def forward(ctx, x, act):
		L.require_gpu(x)
		xc = x.contiguous()
		y = torch.empty_like(xc)
		L.check(L.load().vitk_act_fwd(L.ptr(xc), L.ptr(y), xc.numel(), act, L.dtype_code(xc.dtype), L.stream()))
		ctx.save_for_backward(xc)
		ctx.act = act
		return y

	@staticmethod
	def backward(ctx, dy):
		(xc,) = ctx.saved_tensors
		dyc = cast(dy, xc.dtype)
		dx = torch.empty_like(xc)
		L.check(L.load().vitk_act_bwd(L.ptr(dyc), L.ptr(xc), L.ptr(dx), xc.numel(), ctx.act, L.dtype_code(xc.dtype), L.stream()))
		return dx, None


class ConvTokensFn(torch.autograd.Function):
	"""Conv2d with bias and no normalisation behind it, tokens out: im2col gather + MFMA GEMM with the bias in its epilogue.
	Input NCHW, or token-major [B, H*W, C] with hw = (H, W).  (The patch-merging convolutions between the stages of a
	conv + attention hybrid; a 1x1 convolution on tokens is a plain Linear and needs no gather.)"""

	@staticmethod
	def forward(ctx, x, w, b, cfg):
		ks, stride, pad, hw = cfg
		L.require_gpu(x, w)
		lib = L.load()
		dt = compute_dtype(x)
		xc = cast(x, dt)
		if hw is None:
			Bn, C, H, W = xc.shape
			tokens_in = 0
		else:
			Bn, S, C = xc.shape
			H, W = hw
			assert S == H * W
			tokens_in = 1
		Ho, Wo = _conv_out(H, ks, stride, pad), _conv_out(W, ks, stride, pad)
		Cout, K = w.shape[0], C * ks * ks
		assert K % 8 == 0 or dt == torch.float32, "bf16: C * k * k must be a multiple of 8"
		dtc = L.dtype_code(dt)
		cols = torch.empty((Bn * Ho * Wo, K), dtype=dt, device=x.device)
		L.check(lib.vitk_im2col(L.ptr(xc), L.ptr(cols), Bn, C, H, W, ks, stride, pad, K, tokens_in, dtc, L.stream()))
		y = linear_fwd(cols, weight_as(w, dt).reshape(Cout, K), f32(b))
		ctx.save_for_backward(cols, w)
		ctx.meta = (x.shape, x.dtype, b, cfg, dt, (Bn, C, H, W, tokens_in))
		return y.reshape(Bn, Ho * Wo, Cout)

	@staticmethod
	def backward(ctx, dy):
		cols, w = ctx.saved_tensors
		xshape, xdtype, b, (ks, stride, pad, hw), dt, (Bn, C, H, W, tokens_in) = ctx.meta
		Cout = w.shape[0]
		dy2 = _flat(cast(dy, dt))
		dw, db = linear_wgrad(dy2, cols, bias_grad=b is not None)
		dx = None
		if ctx.needs_input_grad[0]:
			dcols = linear_dgrad(dy2, weight_as(w, dt).reshape(Cout, -1))
			dx = torch.empty(xshape, dtype=dt, device=dy.device)
			L.check(L.load().vitk_col2im(L.ptr(dcols), L.ptr(dx), Bn, C, H, W, ks, stride, pad, dcols.shape[1], tokens_in, L.dtype_code(dt), L.stream()))
			if dx.dtype != xdtype:
				dx = cast(dx, xdtype)
		return dx, grad_as(dw.reshape(w.shape), w), grad_as(db, b), None


class MeanPoolFn(torch.autograd.Function):
	"""x.mean(dim=1) over tokens (models/cvt.py:105 of the reference)."""

	@staticmethod
	def forward(ctx, x):
		L.require_gpu(x)
		xc = x.contiguous()
		Bn, S, d = xc.shape
		y = torch.empty((Bn, d), dtype=xc.dtype, device=xc.device)
		L.check(L.load().vitk_mean_pool_fwd(L.ptr(xc), L.ptr(y), Bn, S, d, L.dtype_code(xc.dtype), L.stream()))
		ctx.meta = (Bn, S, d)
		return y

	@staticmethod
	def backward(ctx, dy):
		Bn, S, d = ctx.meta
		dyc = dy.contiguous()
		dx = torch.empty((Bn, S, d), dtype=dyc.dtype, device=dyc.device)
		L.check(L.load().vitk_mean_pool_bwd(L.ptr(dyc), L.ptr(dx), Bn, S, d, L.dtype_code(dyc.dtype), L.stream()))
		return dx
