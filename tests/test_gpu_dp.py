"""GPU: the data-parallel step on the real kernels, with the collective replaced by a recorder (one process, one
device -- the collective itself is covered over gloo on the CPU in test_dp_gloo.py).  What is checked is the part that
only exists on the GPU path: gradients that the fused block backward adds straight into the flat buffer (gradient sinks)
report their arrival, every bucket is all-reduced exactly once, while backward is still running, and never before the
last kernel that writes into it."""
import pytest
import torch
import torch.distributed as dist
import torch.nn as nn

pytestmark = pytest.mark.gpu

CFG = dict(d_model=64, n_head=1, n_block=3, n_patch=4, d_ff=128, out_d=10, chw=(3, 16, 16))


def _model(dev, layer_scale=False):
	from vit_amd.layers import MHA
	from vit_amd.models import ViT

	class Cfg:
		attention = MHA
		norm = nn.LayerNorm
		activation = "gelu"
		use_layer_scale = layer_scale
		use_qk_norm = False
		dropout = 0.0
	for k, v in CFG.items():
		setattr(Cfg, k, v)
	torch.manual_seed(0)
	model = ViT(Cfg()).to(dev)
	if layer_scale:  # (the initial 1e-4 would leave every gradient behind it at the float32 noise floor)
		with torch.no_grad():
			for n, p in model.named_parameters():
				if n.endswith("layer_scale"):
					p.fill_(0.5)
	return model


def _cvt(dev):
	from conftest import load_golden
	from vit_amd.layers import RMSNorm
	from vit_amd.models import CvT
	_, meta = load_golden("cvt_small_train")

	class Cfg:
		pass
	for k, v in meta["config"].items():
		setattr(Cfg, k, tuple(v) if isinstance(v, list) else v)
	Cfg.norm = RMSNorm
	Cfg.dropout = 0.0
	torch.manual_seed(0)
	return CvT(Cfg()).to(dev).train()


# models whose parameters reach the gradient buffer in different ways: all through sink-aware backward functions (vit);
# LayerScale vectors through autograd's own accumulation, whose hooks fire with the main stream current right behind a
# block whose weight-gradient products were just handed to the second stream (the reference's default configuration has
# use_layer_scale = True, config.py:10-34 there); conv-embedding / depthwise / BatchNorm parameters likewise (cvt)
VARIANTS = {
	"vit": (lambda dev: _model(dev), (8, 3, 16, 16)),
	"vit_layerscale": (lambda dev: _model(dev, layer_scale=True), (8, 3, 16, 16)),
	"cvt": (_cvt, (8, 3, 32, 32)),
}


class _Done:
	def wait(self):
		return True


@pytest.mark.parametrize("variant", ["vit", "vit_layerscale", "cvt"])
@pytest.mark.parametrize("second_stream", [False, True])
@pytest.mark.parametrize("bucket_bytes", [1 << 30, 64 << 10, 4 << 10])
def test_buckets_reduced_once_after_their_last_write(gpu, bucket_bytes, second_stream, variant, monkeypatch):
	"""second_stream: every weight-gradient GEMM of the fused blocks runs on the second stream (size threshold
	off); a collective is ordered behind the stream that is current when it is launched, so the recorder waits for that
	stream only and must still see every gradient write of its bucket."""
	from vit_amd import dp, ops
	records = []
	for lane in ops._wgrad_lanes.values():  # (no backward is running: nothing may be pending on the second stream)
		assert not lane.hooked and not lane.held, (lane.hooked, len(lane.held))
	monkeypatch.setattr(ops, "WGRAD_OVERLAP", second_stream)
	monkeypatch.setattr(ops, "WGRAD_OVERLAP_MIN_FLOPS", 0.0)

	def fake_all_reduce(t, op=None, group=None, async_op=False):
		torch.cuda.current_stream().synchronize()  # what is ordered before a collective launched here: this is what it would read
		records.append((t.data_ptr(), t.numel(), t.detach().clone()))
		t.mul_(2.0)  # "sum over two identical ranks"
		return _Done()

	monkeypatch.setattr(dist, "is_initialized", lambda: True)
	monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
	monkeypatch.setattr(dist, "broadcast", lambda *a, **k: None)
	monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
	ops.clear_grad_sinks()
	make, xshape = VARIANTS[variant]
	model = make(gpu)
	trainer = dp.Trainer(model, lr=1e-2, weight_decay=0.05, bucket_bytes=bucket_bytes)
	assert trainer.sync.world == 2
	g = torch.Generator().manual_seed(7)
	x = torch.randn(*xshape, generator=g).to(gpu)
	y = torch.randint(0, 10, (xshape[0],), generator=g).to(gpu)
	base = trainer.flat.grad.data_ptr()
	for step in range(3):  # the first backward only learns how often each gradient is written; from the second on buckets go out early
		records.clear()
		trainer.zero_grad()
		with torch.autocast("cuda", dtype=torch.bfloat16):
			loss = ops.cross_entropy(model(x), y)
		sections = ops.wgrad_sections()
		loss.backward()
		assert (ops.wgrad_sections() > sections) == second_stream
		during_backward = len(records)
		arrivals = list(trainer.sync.arrivals)
		trainer.sync.finish()
		torch.cuda.synchronize()
		n_buckets = len(trainer.flat.buckets)
		assert len(records) == n_buckets
		if step == 0:
			assert during_backward == 0
		else:
			assert during_backward >= n_buckets - 1
			assert sorted(arrivals) == list(range(len(trainer.flat.params)))  # every parameter, once
			assert trainer.sync.launched == [] and [r[0] for r in records] == [base + 4 * s for s, _, _ in trainer.flat.buckets]  # in bucket order
		spans = sorted(((p - base) // 4, (p - base) // 4 + n) for p, n, _ in records)
		assert spans == sorted((s, e) for s, e, _ in trainer.flat.buckets)
		for p, n, snap in records:
			s = (p - base) // 4
			# nothing was written into the slice after its collective started
			assert torch.equal(trainer.flat.grad[s:s + n], 2.0 * snap), f"bucket at {s} was reduced before its last gradient write"
		trainer.opt.step(grad_scale=0.5)
	# and the reduced gradient is the gradient: compare with a plain run (no trainer, no sinks)
	trainer.close()
	ops.clear_grad_sinks()
	ref = make(gpu)
	with torch.autocast("cuda", dtype=torch.bfloat16):
		ops.cross_entropy(ref(x), y).backward()
	monkeypatch.undo()
	ops.clear_grad_sinks()
	model2 = make(gpu)
	tr2 = dp.Trainer(model2)
	tr2.zero_grad()
	with torch.autocast("cuda", dtype=torch.bfloat16):
		ops.cross_entropy(model2(x), y).backward()
	from conftest import relerr
	for (n, p), (_, q) in zip(ref.named_parameters(), model2.named_parameters()):
		assert relerr(q.grad, p.grad.cpu().double()) < (1e-4 if variant == "vit" else 2e-3), n
	tr2.close()
	ops.clear_grad_sinks()


def test_checkpoint_resume_continues_the_run(gpu, tmp_path):
	"""Save after two steps in the reference's checkpoint format (train.py:121-157 there), load into a fresh model +
	trainer, take a third step on both: the loaded state is bit-identical, the step after it agrees to float
	rounding (the norm-parameter gradients are summed with float atomics, whose order varies between runs).  The optimizer entry of the file also
	loads into a stock torch.optim.AdamW with the reference's parameter groups."""
	from vit_amd import checkpoint as ck, dp, ops
	ops.clear_grad_sinks()
	g = torch.Generator().manual_seed(3)
	x = torch.randn(8, 3, 16, 16, generator=g).to(gpu)
	y = torch.randint(0, 10, (8,), generator=g).to(gpu)

	def one_step(model, tr):
		tr.zero_grad()
		with torch.autocast("cuda", dtype=torch.bfloat16):
			ops.cross_entropy(model(x), y).backward()
		tr.step()

	a = _model(gpu)
	ta = dp.Trainer(a, lr=1e-2, weight_decay=0.05)
	one_step(a, ta)
	one_step(a, ta)
	path = str(tmp_path / "run.pt")
	ck.save_checkpoint(path, a, ta, epoch=4, best_val_loss=1.5)
	b = _model(gpu)
	with torch.no_grad():
		for p in b.parameters():
			p.add_(1.0)  # make sure the load is what sets them
	tb = dp.Trainer(b, lr=1.0, weight_decay=0.0)
	epoch, best = ck.load_checkpoint(path, b, tb, map_location=gpu)
	assert (epoch, best) == (5, 1.5) and tb.opt.step_count == 2 and tb.opt.lr == 1e-2 and tb.opt.weight_decay == 0.05
	assert torch.equal(tb.flat.param, ta.flat.param) and torch.equal(tb.opt.m, ta.opt.m) and torch.equal(tb.opt.v, ta.opt.v)
	assert torch.equal(tb.opt.shadow, ta.opt.shadow)  # (the transposed copies are covered by the next step: the input-gradient GEMMs read them)
	one_step(a, ta)
	one_step(b, tb)
	assert tb.opt.step_count == 3
	for got, want in ((tb.flat.param, ta.flat.param), (tb.opt.m, ta.opt.m), (tb.opt.v, ta.opt.v)):
		assert torch.allclose(got, want, rtol=1e-4, atol=1e-7), float((got - want).abs().max())
	# the same file feeds a stock AdamW over the reference's two groups
	sd = torch.load(path, weights_only=True)
	c = _model("cpu")
	c.load_state_dict(sd["model_state_dict"])
	dec = [p for n, p in c.named_parameters() if not dp.no_decay(n)]
	nod = [p for n, p in c.named_parameters() if dp.no_decay(n)]
	opt = torch.optim.AdamW([{"params": dec, "weight_decay": 0.05}, {"params": nod, "weight_decay": 0.0}], lr=1e-2)
	opt.load_state_dict(sd["optimizer_state_dict"])
	assert all(float(s["step"]) == 2.0 for s in opt.state.values()) and len(opt.state) == len(dec) + len(nod)
	ops.clear_grad_sinks()


def test_module_by_module_functions_write_their_gradient_sinks(gpu):
	"""CvT with its blocks run module by module (Linear, norm, feed-forward and conv-attention functions each on their own;
	the one-node-per-block form is compared with it at the end) and a ViT with
	LayerScale + RMSNorm: under a Trainer the backward functions add parameter gradients straight into the flat buffer;
	the result must equal the gradients plain autograd accumulates without sinks, and a second backward must add to it."""
	from conftest import load_golden, relerr
	from vit_amd import dp, ops
	from vit_amd.layers import RMSNorm
	from vit_amd.models import CvT
	_, meta = load_golden("cvt_small_train")

	def make():
		class Cfg:
			pass
		for k, v in meta["config"].items():
			setattr(Cfg, k, tuple(v) if isinstance(v, list) else v)
		Cfg.norm = RMSNorm
		Cfg.dropout = 0.0
		torch.manual_seed(0)
		return CvT(Cfg()).to(gpu).train()
	g = torch.Generator().manual_seed(5)
	x = torch.randn(4, 3, 32, 32, generator=g).to(gpu)
	y = torch.randint(0, 10, (4,), generator=g).to(gpu)

	def backward(model):
		with torch.autocast("cuda", dtype=torch.bfloat16):
			ops.cross_entropy(model(x), y).backward()
	import vit_amd.layers.cvt as cvt_layers
	ops.clear_grad_sinks()
	cvt_layers.FUSED_BLOCK = False
	try:
		plain = make()
		backward(plain)
		want = {k: p.grad.clone() for k, p in plain.named_parameters()}
		sunk = make()
		tr = dp.Trainer(sunk)
		tr.zero_grad()
		backward(sunk)
	finally:
		cvt_layers.FUSED_BLOCK = True
	# one autograd node per block (what runs by default): same gradients, with and without sinks
	fused = make()
	backward(fused)
	for k, p in fused.named_parameters():
		# (the residual adds move into GEMM epilogues and norm kernels, so bf16 roundings differ along the whole chain; the bias
		# gradients -- sums of many small terms of both signs -- show it most: measured up to 4.9e-2)
		assert relerr(p.grad, want[k]) < 8e-2, k
	fused_sunk = make()
	tr2 = dp.Trainer(fused_sunk)
	tr2.zero_grad()
	backward(fused_sunk)
	for k, p in fused_sunk.named_parameters():
		assert relerr(p.grad, fused.get_parameter(k).grad) < 2e-3, k
	tr2.close()
	assert sorted(tr.flat.names) == sorted(k for k, _ in sunk.named_parameters())
	for k, p, o in zip(tr.flat.names, tr.flat.params, tr.flat.offsets):
		assert p.grad.data_ptr() == tr.flat.grad.data_ptr() + 4 * o, k  # still the view of the flat buffer
		assert relerr(p.grad, want[k]) < 2e-3, k
	# the three projections of a conv-attention module are ONE matrix in the flat buffers (q | k | v weights, then biases)
	at = tr.flat.names.index("stage1_blocks.0.attn.q_proj.weight")
	assert tr.flat.names[at:at + 8] == ["stage1_blocks.0.attn." + s for s in ("q_proj.weight", "k_proj.weight", "v_proj.weight", "q_proj.bias", "k_proj.bias",
		"v_proj.bias", "conv_proj_k.weight", "conv_proj_v.weight")]
	blk = sunk.stage1_blocks[0].attn
	assert ops._joint([blk.q_proj.weight, blk.k_proj.weight, blk.v_proj.weight]) is not None
	assert ops._joint([blk.q_proj.weight.grad, blk.k_proj.weight.grad, blk.v_proj.weight.grad]) is not None
	assert [tr.flat.names[i] for i in tr.flat.module_order] == [k for k, _ in sunk.named_parameters()]
	sunk.load_state_dict(plain.state_dict())  # BatchNorm statistics moved with the first forward
	backward(sunk)  # no zero_grad: gradients accumulate, as .grad does
	for k, p in sunk.named_parameters():
		assert relerr(p.grad, 2 * want[k]) < 5e-2, k
	ops.clear_grad_sinks()


def test_trainer_steps_like_torch_adamw_with_the_reference_groups(gpu):
	"""dp.Trainer (flat buffers, fused device-side AdamW, weight-decay mask from the reference's name rule, bf16 and
	transposed weight copies refreshed per step) against torch.optim.AdamW over get_param_groups (utils.py:102-121,
	main.py:100 there) on a twin model: same float32 forward / backward kernels, so only the optimizer differs."""
	from conftest import relerr
	from vit_amd import dp, ops
	ops.clear_grad_sinks()
	g = torch.Generator().manual_seed(21)
	xs = [torch.randn(8, 3, 16, 16, generator=g).to(gpu) for _ in range(4)]
	ys = [torch.randint(0, 10, (8,), generator=g).to(gpu) for _ in range(4)]
	a, b = _model(gpu), _model(gpu)
	dec = [p for n, p in b.named_parameters() if not (n.endswith("bias") or "norm" in n.lower() or "layerscale" in n.lower())]
	nod = [p for n, p in b.named_parameters() if (n.endswith("bias") or "norm" in n.lower() or "layerscale" in n.lower())]
	assert dec and nod
	opt = torch.optim.AdamW([{"params": dec, "weight_decay": 0.05}, {"params": nod, "weight_decay": 0.0}], lr=3e-3, betas=(0.9, 0.999), eps=1e-8)
	tr = dp.Trainer(a, lr=3e-3, weight_decay=0.05)
	for x, y in zip(xs, ys):
		tr.zero_grad()
		ops.cross_entropy(a(x), y).backward()
		tr.step()
		opt.zero_grad()
		ops.cross_entropy(b(x), y).backward()
		opt.step()
		ops.invalidate_casts()
		ops.weights_changed()  # the stock optimizer wrote b's weights behind the kernels' cached copies
		for (n, p), q in zip(a.named_parameters(), b.parameters()):
			assert relerr(p, q) < 2e-5, n
	# the bf16 copy the next forward reads is the rounded float32 parameter
	assert torch.equal(tr.opt.shadow, tr.flat.param.to(torch.bfloat16))
	ops.clear_grad_sinks()


def test_weight_gradients_on_the_second_stream_are_the_same_gradients(gpu):
	"""The fused block's weight-gradient GEMMs issued on the second stream (behind the input-gradient launch
	that shares their dy, joined at the end of backward) against the same step with everything on one stream: the kernels
	and their summation orders are the same, so the gradients of the blocks' Linear layers must be identical bit for bit,
	also when a second backward accumulates into them (the norm and patch-embedding gradients end in float atomics and
	differ in the last bit from run to run on one stream already), and the optimizer step that follows must read complete
	gradients."""
	from vit_amd import dp, ops
	ops.clear_grad_sinks()
	g = torch.Generator().manual_seed(11)
	x = torch.randn(32, 3, 16, 16, generator=g).to(gpu)
	y = torch.randint(0, 10, (32,), generator=g).to(gpu)
	saved = ops.WGRAD_OVERLAP, ops.WGRAD_OVERLAP_MIN_FLOPS
	out = {}
	try:
		for mode in (False, True):
			ops.WGRAD_OVERLAP, ops.WGRAD_OVERLAP_MIN_FLOPS = mode, 0.0
			ops.clear_grad_sinks()
			model = _model(gpu)
			tr = dp.Trainer(model, lr=1e-2, weight_decay=0.05)
			n0 = ops.wgrad_sections()
			grads = []
			for _ in range(2):
				tr.zero_grad()
				with torch.autocast("cuda", dtype=torch.bfloat16):
					ops.cross_entropy(model(x), y).backward()
					ops.cross_entropy(model(x.flip(0)), y).backward()  # accumulates
				grads.append(tr.flat.grad.clone())  # (on the main stream: backward has joined the second stream)
				tr.step()
			# per backward: 4 products per full block, 3 in the class-rows form of the last block (fc_out, fc_in, key / value projection)
			assert (ops.wgrad_sections() - n0 == 2 * 2 * (4 * (CFG["n_block"] - 1) + 3)) if mode else (ops.wgrad_sections() == n0)
			out[mode] = (grads, tr.flat.param.clone())
			spans = [(n, o, p.numel()) for n, o, p in zip(tr.flat.names, tr.flat.offsets, tr.flat.params)]
			tr.close()
	finally:
		ops.WGRAD_OVERLAP, ops.WGRAD_OVERLAP_MIN_FLOPS = saved
		ops.clear_grad_sinks()
	exact = 0
	for a, b in zip(out[False][0], out[True][0]):
		for n, o, k in spans:
			if n.startswith("blocks.") and "norm" not in n:
				assert torch.equal(a[o:o + k], b[o:o + k]), n
				exact += 1
			else:
				assert float((a[o:o + k] - b[o:o + k]).abs().max()) <= 1e-6 * float(a[o:o + k].abs().max()) + 1e-12, n
	assert exact == 2 * 7 * CFG["n_block"]  # qkv, out (+bias), fc_in (+bias), fc_out (+bias) per block and step
	assert float((out[False][1] - out[True][1]).abs().max()) < 1e-6


@pytest.mark.parametrize("variant", ["vit", "vit_layerscale", "cvt"])
@pytest.mark.parametrize("flat", [False, True])
def test_weight_gradients_launched_together_are_the_same_gradients(gpu, variant, flat, monkeypatch):
	"""The weight-gradient products of a block leave in one grouped call at the end of its backward (ops.wgrad_group);
	one launch per product gives the same gradients up to the float32 summation order.  Enough tokens (1088 / 4096 ...)
	that the grouped kernel really takes them; flat: into the gradient sinks of a dp.FlatModel."""
	from conftest import relerr
	from vit_amd import dp, ops
	monkeypatch.setattr(ops, "WGRAD_OVERLAP", False)
	make, xshape = VARIANTS[variant]
	g = torch.Generator().manual_seed(11)
	x = torch.randn(64, *xshape[1:], generator=g).to(gpu)
	y = torch.randint(0, 10, (64,), generator=g).to(gpu)
	grads = {}
	for grouped in (False, True):
		monkeypatch.setattr(ops, "WGRAD_GROUP", grouped)
		ops.clear_grad_sinks()
		model = make(gpu)
		trainer = dp.Trainer(model) if flat else None
		if trainer is not None:
			trainer.zero_grad()
		before = ops.GROUP_CALLS
		with torch.autocast("cuda", dtype=torch.bfloat16):
			ops.cross_entropy(model(x), y).backward()
		torch.cuda.synchronize()
		assert (ops.GROUP_CALLS > before) == grouped
		grads[grouped] = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()}
		if trainer is not None:
			trainer.close()
		ops.clear_grad_sinks()
	for n, gref in grads[False].items():
		assert relerr(grads[True][n], gref) < 2e-5, n


def test_weight_gradient_on_the_second_stream_outside_backward(gpu):
	"""ops.linear_wgrad(overlap=True) called outside a backward pass joins the second stream before it returns."""
	from vit_amd import ops
	saved = ops.WGRAD_OVERLAP, ops.WGRAD_OVERLAP_MIN_FLOPS
	try:
		ops.WGRAD_OVERLAP, ops.WGRAD_OVERLAP_MIN_FLOPS = True, 0.0
		g = torch.Generator().manual_seed(2)
		dy = torch.randn(4096, 256, generator=g).to(gpu).bfloat16()
		xx = torch.randn(4096, 128, generator=g).to(gpu).bfloat16()
		dw = torch.zeros(256, 128, device=gpu)
		db = torch.zeros(256, device=gpu)
		n0 = ops.wgrad_sections()
		ops.linear_wgrad(dy, xx, bias_grad=True, dw_out=dw, db_out=db, overlap=True)
		assert ops.wgrad_sections() == n0 + 1
		got = dw.clone()  # main stream
		want = dy.double().t() @ xx.double()
		assert float((got.double() - want).abs().max() / want.abs().max()) < 1e-5
		assert float((db.double() - dy.double().sum(0)).abs().max()) < 1e-2
	finally:
		ops.WGRAD_OVERLAP, ops.WGRAD_OVERLAP_MIN_FLOPS = saved


def test_pipelined_optimizer_step_is_the_same_training_run(gpu):
	"""dp.Trainer(pipelined_step=True): the optimizer runs piece by piece on the second stream while the next forward has
	already started; every kernel that reads a parameter waits for its piece, the step clears the gradients it consumed.
	Same kernels on the same numbers as the plain step: parameters, moments and the bf16 / transposed copies after five
	steps agree to the run-to-run noise of the norm gradients' float atomics, and a checkpoint taken in between sees the
	finished step."""
	from vit_amd import dp, ops
	g = torch.Generator().manual_seed(5)
	xs = [torch.randn(16, 3, 16, 16, generator=g).to(gpu) for _ in range(5)]
	ys = [torch.randint(0, 10, (16,), generator=g).to(gpu) for _ in range(5)]
	out = {}
	for mode in (False, True):
		ops.clear_grad_sinks()
		model = _model(gpu)
		# (a small learning rate: the run-to-run noise of the norm gradients' float atomics, amplified through the bf16 rounding
		# of the weights, moves the losses of two plain runs apart by 1e-3 within five steps at lr 3e-3)
		tr = dp.Trainer(model, lr=1e-4, weight_decay=0.05, pipelined_step=mode)
		assert len(tr.opt.pieces) == 3 + CFG["n_block"] + 2 - 1 or len(tr.opt.pieces) >= CFG["n_block"] + 2
		losses = []
		for i, (x, y) in enumerate(zip(xs, ys)):
			tr.zero_grad()
			with torch.autocast("cuda", dtype=torch.bfloat16):
				loss = ops.cross_entropy(model(x), y)
			loss.backward()
			tr.step()
			losses.append(loss.detach())
			if i == 2:
				sd = tr.opt.state_dict()  # waits for the step by itself
				assert int(tr.opt.step_state[0]) == 3
				assert all(float(st["step"]) == 3.0 for st in sd["state"].values())
		if mode:
			assert tr.opt.grads_cleared
		tr.wait_step()
		if mode:
			assert float(tr.flat.grad.abs().max()) == 0.0  # consumed and cleared
		two_d = torch.zeros(tr.flat.numel, dtype=torch.bool, device=gpu)  # (the transposed copies exist for the matrices only)
		for p_, o in zip(tr.flat.params, tr.flat.offsets):
			if p_.dim() == 2:
				two_d[o:o + p_.numel()] = True
		out[mode] = (tr.flat.param.clone(), tr.opt.m.clone(), tr.opt.v.clone(), tr.opt.shadow.clone(), tr.opt.shadow_t[two_d].clone(), torch.stack(losses).cpu())
		assert torch.equal(tr.opt.shadow, tr.flat.param.to(torch.bfloat16))
		tr.close()
	ops.clear_grad_sinks()
	a, b = out[False], out[True]
	assert float((a[5] - b[5]).abs().max()) < 1e-3, (a[5], b[5])
	for i in range(3):
		assert float((a[i] - b[i]).abs().max()) <= 1e-5 * float(a[i].abs().max()), i
	assert float((a[3].float() - b[3].float()).abs().max()) <= 1e-2 * float(a[3].float().abs().max())
	assert float((a[4].float() - b[4].float()).abs().max()) <= 1e-2 * float(a[4].float().abs().max())


def test_pipelined_step_hands_over_to_plain_steps_and_to_graph_capture(gpu):
	"""The pipelined step clears the gradients itself and tells the next zero_grad() to skip its fill.  A plain step after
	it leaves the gradients alone, so the zero_grad() behind THAT one must fill again; and a step captured into a HIP graph
	from a trainer that was running pipelined steps must contain its own zero fill (replays do not accumulate)."""
	from vit_amd import dp, ops
	ops.clear_grad_sinks()
	g = torch.Generator().manual_seed(9)
	x = torch.randn(16, 3, 16, 16, generator=g).to(gpu)
	y = torch.randint(0, 10, (16,), generator=g).to(gpu)
	model = _model(gpu)
	tr = dp.Trainer(model, lr=0.0, weight_decay=0.0, pipelined_step=True)  # (parameters stay put: equal inputs give equal gradients)

	def loss_fn(xi, yi):
		with torch.autocast("cuda", dtype=torch.bfloat16):
			return ops.cross_entropy(model(xi), yi)
	tr.zero_grad()
	loss_fn(x, y).backward()
	tr.step()
	assert tr.opt.grads_cleared
	tr.pipelined_step = False
	loss_fn(x, y).backward()  # (into the gradients the pipelined step cleared)
	tr.wait_step()
	once = tr.flat.grad.clone()
	tr.step()
	assert not tr.opt.grads_cleared and torch.equal(tr.flat.grad, once)  # the plain step leaves them
	tr.zero_grad()
	assert float(tr.flat.grad.abs().max()) == 0.0
	# capture from a trainer in the middle of pipelined steps
	tr.pipelined_step = True
	loss_fn(x, y).backward()
	tr.step()
	gs = dp.GraphedStep(tr, loss_fn, [x, y])
	assert not tr.pipelined_step
	gs(x, y)
	torch.cuda.synchronize()
	first = tr.flat.grad.clone()
	gs(x, y)
	torch.cuda.synchronize()
	assert float((tr.flat.grad - first).abs().max()) <= 1e-6 * float(first.abs().max())  # not twice the gradient
	assert float((first - once).abs().max()) <= 1e-6 * float(once.abs().max())
	gs.close()
	tr.close()
	ops.clear_grad_sinks()


def test_forward_after_optimizer_steps_reads_the_updated_conv_stem(gpu):
	"""The 7x7 conv embedding's weight (K = 147) is read through a K-padded bf16 copy that is cached between forward passes;
	the fused optimizer rewrites the parameter without touching its version counter, so the cache must follow the
	optimizer's own epoch: after two steps the model's forward must equal that of a fresh model holding the same
	parameters (whose caches are empty)."""
	from conftest import load_golden, relerr
	from vit_amd import dp, ops
	from vit_amd.layers import RMSNorm
	from vit_amd.models import CvT
	_, meta = load_golden("cvt_small_train")

	def make():
		class Cfg:
			pass
		for k, v in meta["config"].items():
			setattr(Cfg, k, tuple(v) if isinstance(v, list) else v)
		Cfg.norm = RMSNorm
		Cfg.dropout = 0.0
		torch.manual_seed(0)
		return CvT(Cfg()).to(gpu).train()
	g = torch.Generator().manual_seed(5)
	x = torch.randn(8, 3, 32, 32, generator=g).to(gpu)
	y = torch.randint(0, 10, (8,), generator=g).to(gpu)
	ops.clear_grad_sinks()
	model = make()
	tr = dp.Trainer(model, lr=3e-2, weight_decay=0.0)
	with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
		before = model(x).float()
	for _ in range(2):
		tr.zero_grad()
		with torch.autocast("cuda", dtype=torch.bfloat16):
			ops.cross_entropy(model(x), y).backward()
		tr.step()
	with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
		after = model(x).float()
	sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
	tr.close()
	ops.clear_grad_sinks()
	fresh = make()
	fresh.load_state_dict(sd)
	ops.invalidate_casts()
	with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
		want = fresh(x).float()
	assert relerr(after, before) > 5e-2  # the steps did move the model
	assert relerr(after, want) < 2e-2


def test_second_stream_orders_a_long_kernel_behind_run_behind_and_the_end_of_backward(gpu):
	"""Stream ordering of the second (weight-gradient) stream, with a deliberately long chain of kernels on it (~20 ms of
	8192^3 GEMMs, then a flag write): (1) a callback issued through run_behind -- the way a bucket's collective is
	launched -- reads the flag with the second stream current and must see it written; (2) autograd's end-of-backward
	callback joins the stream, so a read on the main stream right after backward() must see it too.  The stream is one
	of torch's pool streams (an externally created HIP stream wrapped as ExternalStream let such waits return early or
	never in a process that had captured graphs before)."""
	from vit_amd import ops
	lane = ops.second_stream(gpu)
	big = (torch.randn(8192, 8192, generator=torch.Generator().manual_seed(1)) * 0.01).to(torch.bfloat16).to(gpu)
	out = torch.empty_like(big)
	flag = torch.zeros(4, device=gpu)
	seen = {}

	class Fn(torch.autograd.Function):
		@staticmethod
		def forward(ctx, t):
			return t.clone()

		@staticmethod
		def backward(ctx, g):
			with lane.section(None, big, out):
				for _ in range(24):
					ops.gemm(big, big, out, 8192, 8192, 8192, 8192, 8192, 8192)
				flag.fill_(1.0)
			assert lane.hooked  # the join is queued for the end of this backward pass
			seen["unordered"] = flag.clone()  # main stream, no join yet: not ordered behind the chain (recorded, not asserted)
			lane.run_behind(lambda: seen.__setitem__("behind", flag.clone()))
			return g

	for rep in range(3):
		flag.zero_()
		torch.cuda.synchronize()
		t = torch.ones(8, device=gpu, requires_grad=True)
		Fn.apply(t).sum().backward()
		seen["after"] = flag.clone()  # main stream, after backward(): the end-of-backward callback has joined the second stream
		torch.cuda.synchronize()
		assert not lane.hooked and not lane.held
		assert float(seen["behind"][0]) == 1.0, (rep, "a launch made through run_behind ran ahead of the second stream")
		assert float(seen["after"][0]) == 1.0, (rep, "the main stream read a result of the second stream before it was written")
		assert float(out.float().abs().max()) > 0.0


def test_graphed_step_with_gradient_sync_outside_the_graph(gpu, monkeypatch):
	"""dp.GraphedStep on a trainer whose gradient synchronisation is active (more than one rank): the graph holds forward +
	backward, the bucket all-reduces and the optimizer step run eagerly behind every replay.  The collective is replaced by
	"two identical ranks" (the sum is twice the local gradient, the optimizer applies 1 / world), so the run must follow a
	single-process captured run on the same data; every bucket is reduced once per step, in order, outside the capture."""
	from vit_amd import dp, ops
	g = torch.Generator().manual_seed(13)
	xs = [torch.randn(16, 3, 16, 16, generator=g).to(gpu) for _ in range(4)]
	ys = [torch.randint(0, 10, (16,), generator=g).to(gpu) for _ in range(4)]

	def run(sharded):
		ops.clear_grad_sinks()
		calls = []
		if sharded:
			def fake_all_reduce(t, op=None, group=None, async_op=False):
				assert not torch.cuda.is_current_stream_capturing()
				calls.append(t.data_ptr())
				t.mul_(2.0)
				return _Done()
			monkeypatch.setattr(dist, "is_initialized", lambda: True)
			monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
			monkeypatch.setattr(dist, "broadcast", lambda *a, **k: None)
			monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
		model = _model(gpu)
		tr = dp.Trainer(model, lr=1e-3, weight_decay=0.05, bucket_bytes=64 << 10)
		assert tr.sync.active == sharded

		def loss_fn(xi, yi):
			with torch.autocast("cuda", dtype=torch.bfloat16):
				return ops.cross_entropy(model(xi), yi)
		gs = dp.GraphedStep(tr, loss_fn, [xs[0], ys[0]])
		calls.clear()
		losses = []
		for x, y in zip(xs, ys):
			losses.append(gs(x, y).clone())
		torch.cuda.synchronize()
		nb = len(tr.flat.buckets)
		if sharded:
			base = tr.flat.grad.data_ptr()
			assert calls == [base + 4 * s for s, _, _ in tr.flat.buckets] * len(xs)
			assert tr.opt.step_count == len(xs)
		out = tr.flat.param.clone(), torch.stack(losses).cpu(), nb
		gs.close()
		tr.close()
		monkeypatch.undo()
		ops.clear_grad_sinks()
		return out
	p1, l1, _ = run(False)
	p2, l2, nb = run(True)
	assert nb > 1
	assert float((l1 - l2).abs().max()) < 1e-3, (l1, l2)
	assert float((p1 - p2).abs().max()) <= 1e-4 * float(p1.abs().max())
