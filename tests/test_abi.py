"""CPU: the C-ABI library loads and exports every symbol include/vitk.h declares, the ctypes table matches the
header, and the drop-in modules keep the reference's parameter names and refuse CPU tensors."""
import os
import re

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
	src = open(os.path.join(ROOT, "include", "vitk.h")).read()
	src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
	return sorted(set(re.findall(r"\b(vitk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
	import vit_amd._lib as L
	lib = L.load()
	names = header_functions()
	assert len(names) >= 30
	for n in names:
		assert hasattr(lib, n), f"{n} declared in include/vitk.h but not exported by libvitk.so"
	assert sorted(L.SIGNATURES) == names, "ctypes signature table and header disagree"
	assert lib.vitk_version() == 100
	assert b"null" in lib.vitk_strerror(-4)


def test_gemm_desc_layout_matches_header():
	import ctypes
	import vit_amd._lib as L
	# 3 pointers, 6 int64, 12 int32, 3 pointers, pointer + int64, pointer + int64
	# ... then 2 floats + 2 uint64 + 1 int64 (fused dropout)
	assert ctypes.sizeof(L.GemmDesc) == 3 * 8 + 6 * 8 + 12 * 4 + 3 * 8 + 16 + 16 + 8 + 16 + 8 + 16 + 8
	assert L.GemmDesc.drop_seed.offset == ctypes.sizeof(L.GemmDesc) - 24
	assert L.GemmDesc.bias.offset == 3 * 8 + 6 * 8 + 12 * 4


def test_argument_validation_without_gpu():
	"""Host-side validation returns error codes before any launch (no GPU needed)."""
	import vit_amd._lib as L
	lib = L.load()
	assert lib.vitk_gemm(None, None) == -4
	d = L.GemmDesc()
	assert lib.vitk_gemm(d, None) == -4
	assert lib.vitk_layernorm_fwd(None, None, None, None, None, None, None, None, 1, 1, 1e-5, 0, None) == -4
	assert lib.vitk_attention_fwd(None, None, None, None, None, 1, 1, 1, 1, 64, 64, 64, 64, 64, 0.0, 1, None) == -4
	assert lib.vitk_adamw(None, None, None, None, None, None, 1, 0.1, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None) == -4


def test_vit_state_dict_keys_match_reference_goldens():
	from conftest import load_golden, sub
	from vit_amd.layers import MHA, RMSNorm
	from vit_amd.models import ViT
	for name in ("vit_ln_gelu", "vit_rms_swiglu_ls"):
		g, meta = load_golden(name)
		c = meta["config"]

		class Cfg:
			pass
		for k, v in c.items():
			setattr(Cfg, k, tuple(v) if isinstance(v, list) else v)
		Cfg.attention = MHA
		Cfg.norm = nn.LayerNorm if c["norm"] == "layernorm" else RMSNorm
		Cfg.dropout = 0.0
		Cfg.use_qk_norm = False
		m = ViT(Cfg())
		ref = sub(g, "sd")
		assert list(m.state_dict().keys()) == list(ref.keys())
		for k, v in m.state_dict().items():
			assert tuple(v.shape) == tuple(ref[k].shape), k
		m.load_state_dict(ref, strict=True)


def test_cvt_state_dict_keys_match_reference_goldens():
	from conftest import load_golden, sub
	from vit_amd.layers import RMSNorm
	from vit_amd.models import CvT
	g, meta = load_golden("cvt_small_train")

	class Cfg:
		pass
	for k, v in meta["config"].items():
		setattr(Cfg, k, tuple(v) if isinstance(v, list) else v)
	Cfg.norm = RMSNorm
	Cfg.dropout = 0.0
	m = CvT(Cfg())
	ref = sub(g, "sd")
	assert sorted(m.state_dict().keys()) == sorted(ref.keys())
	m.load_state_dict(ref, strict=True)


def test_cpu_tensors_are_rejected():
	from vit_amd.layers import LayerNorm, RMSNorm
	with pytest.raises(RuntimeError, match="no CPU fallback"):
		RMSNorm(8)(torch.randn(2, 8))
	with pytest.raises(RuntimeError, match="no CPU fallback"):
		LayerNorm(8)(torch.randn(2, 8))


def test_feedforward_interface():
	from vit_amd.layers import FeedForward

	class Cfg:
		d_model = 8; d_ff = 16; dropout = 0.0; activation = "swiglu"
	ff = FeedForward(Cfg())
	assert ff.fc_in.weight.shape == (32, 8) and ff.fc_out.weight.shape == (8, 16) and ff.uses_gate
	Cfg.activation = "reglu"
	ff = FeedForward(Cfg())
	assert ff.fc_in.weight.shape == (16, 8) and ff.uses_gate and not ff.gated  # listed as gated, behaves as relu
	Cfg.activation = "tanh"
	with pytest.raises(ValueError):
		FeedForward(Cfg())


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference checkout not present (GPU box)")
def test_reference_models_construct_on_top_of_these_layers():
	"""The import switch (alias this package's layers as `layers`): the reference's own models/vit.py and models/cvt.py build, unchanged, on
	vit_amd.layers (run in a subprocess so that sys.modules of the test process stays clean)."""
	import subprocess
	import sys
	code = r'''
import sys
sys.path.insert(0, %r)
import vit_amd.layers
import vit_amd.layers.embeddings, vit_amd.layers.attention, vit_amd.layers.norm, vit_amd.layers.transformers
import vit_amd.layers.vit, vit_amd.layers.cvt, vit_amd.layers.conv
sys.modules["layers"] = vit_amd.layers
for m in ("embeddings", "attention", "norm", "transformers", "vit", "cvt", "conv"):
	sys.modules["layers." + m] = getattr(vit_amd.layers, m)
sys.path.insert(1, "/root/reference")
import torch.nn as nn
from config import ViTConfig, CvTConfig
from layers.attention import MHA
from models.vit import ViT
from models.cvt import CvT
import vit_amd.models as ours
class C(ViTConfig):
	attention = MHA
	norm = nn.LayerNorm
	dropout = 0.0
ref_on_ours = ViT(C())
assert type(ref_on_ours.blocks[0]).__module__ == "vit_amd.layers.vit"
assert list(ref_on_ours.state_dict()) == list(ours.ViT(C()).state_dict())
class CC(CvTConfig):
	dropout = 0.0
cv = CvT(CC())
assert type(cv.stage1_blocks[0].attn).__module__ == "vit_amd.layers.attention"
assert sorted(cv.state_dict()) == sorted(ours.CvT(CC()).state_dict())
print("ok")
''' % ROOT
	out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
	assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
