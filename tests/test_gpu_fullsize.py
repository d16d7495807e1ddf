"""GPU parity at the headline geometry: the whole ViT-B/16 (d 768, 12 heads, 12 blocks, d_ff 3072, 197 tokens, 1000 classes)
forward + cross entropy + backward in bf16 autocast against the float32 CPU oracle, every parameter gradient included
(bars: 2e-2 on logits and loss, 2.5e-2 on gradients, max|a-b| / max|b| per tensor); and the stand-alone patch gather of
layers.embeddings.patchify against the reference's own output, bit for bit.  Measured errors go to
gpurun_out/parity_errors.json when that directory exists (the table in the design document is built from it)."""
import json
import os

import pytest
import torch
import torch.nn as nn

from conftest import ROOT, load_golden, relerr

pytestmark = pytest.mark.gpu


def _record(section, errs):
	out = os.path.join(ROOT, "gpurun_out")
	if not os.path.isdir(out):
		return
	path = os.path.join(out, "parity_errors.json")
	data = json.load(open(path)) if os.path.exists(path) else {}
	data[section] = errs
	json.dump(data, open(path, "w"), indent=1, sort_keys=True)


def test_vit_b16_full_geometry_against_the_oracle(gpu):
	from oracle import vit_oracle as O
	from vit_amd import _lib as L, ops
	from vit_amd.layers import MHA
	from vit_amd.models import ViT

	class Cfg:
		attention = MHA
		norm = nn.LayerNorm
		activation = "gelu"
		use_layer_scale = False
		use_qk_norm = False
		dropout = 0.0
		d_model, n_head, n_block, n_patch, d_ff, out_d, chw = 768, 12, 12, 14, 3072, 1000, (3, 224, 224)
	B = 32
	ops.clear_grad_sinks()
	torch.manual_seed(0)
	model = ViT(Cfg()).to(gpu)
	g = torch.Generator().manual_seed(11)
	x = torch.randn(B, 3, 224, 224, generator=g)
	y = torch.randint(0, 1000, (B,), generator=g)
	# the paths this size takes: bf16 MFMA GEMMs on 6304 rows (more than one tile per CU for the wide layers, K-split weight
	# gradients over 6304 tokens), input gradients against transposed weight copies
	rows = B * 197
	d = L.GemmDesc()
	d.M, d.N, d.K, d.lda, d.ldb, d.ldc = rows, 3072, 768, 768, 768, 3072
	d.in_dtype = d.out_dtype = L.BF16
	d.A = d.B = d.C = 256  # (only alignment is inspected)
	assert L.load().vitk_gemm_uses_bf16_mfma(d) == 1
	assert ops.DGRAD_TRANSPOSED_WEIGHTS and rows >= 4096
	xg = x.to(gpu).requires_grad_(True)
	with torch.autocast("cuda", dtype=torch.bfloat16):
		logits = model(xg)
		loss = ops.cross_entropy(logits.float(), y.to(gpu))
	loss.backward()
	sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
	fwd = lambda im, s: O.vit_fwd(im, s, 14, 12, 12, "layernorm", "gelu")
	want_logits, want_loss, want_grads, want_gx = O.loss_and_grads(fwd, sd, x, y)
	errs = {"logits": relerr(logits, want_logits), "loss": relerr(loss, want_loss), "grad_x": relerr(xg.grad, want_gx)}
	for k, p in model.named_parameters():
		assert p.grad is not None, k
		errs["grad/" + k] = relerr(p.grad, want_grads[k])
	_record("vit_b16_batch32_bf16", errs)
	assert errs["logits"] < 2e-2 and errs["loss"] < 2e-2
	worst = max(errs, key=errs.get)
	assert errs[worst] < 2.5e-2, (worst, errs[worst])  # measured: 2.0e-2 (a norm weight), everything else <= 1.6e-2
	assert (logits.float().argmax(-1).cpu() == want_logits.argmax(-1)).float().mean() >= 0.9  # bf16: near-ties may flip


@pytest.mark.parametrize("name", ["vit_ln_gelu", "vit_p4_n65", "vit_p16_geglu"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patchify_layer_is_bit_exact(gpu, name, dtype):
	"""layers.embeddings.patchify (what the reference's unchanged models/vit.py:46 calls) is a pure gather."""
	from vit_amd.layers.embeddings import patchify
	g, meta = load_golden(name)
	x = g["x"].to(gpu).to(dtype)
	got = patchify(x, meta["config"]["n_patch"])
	want = g["patches"].to(dtype)
	assert got.shape == want.shape and got.dtype == dtype
	assert torch.equal(got.cpu(), want)
