"""ctypes binding of libvitk.so (the C ABI declared in include/vitk.h).

The library is the only compute path: importing this module never falls back to PyTorch kernels.
If the shared object is missing, every call raises with the build command.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITK_LIB") or os.path.join(_HERE, "libvitk.so")  # VITK_LIB: explicit path of another build (A/B runs)

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_SILU, ACT_RELU, ACT_RELU6 = 0, 1, 2, 3, 4
ACT_CODES = {"none": ACT_NONE, "gelu": ACT_GELU, "silu": ACT_SILU, "relu": ACT_RELU, "relu6": ACT_RELU6}
K_MAJOR, MN_MAJOR = 0, 1
EPI_NONE, EPI_ACT, EPI_RESIDUAL, EPI_ACTGRAD = 0, 1, 2, 3


class GemmDesc(ctypes.Structure):
	"""Mirror of `vitk_gemm_desc` (include/vitk.h)."""
	_fields_ = [
		("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
		("M", c_int64), ("N", c_int64), ("K", c_int64),
		("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64),
		("a_layout", c_int32), ("b_layout", c_int32),
		("in_dtype", c_int32), ("out_dtype", c_int32),
		("epilogue", c_int32), ("act", c_int32),
		("accumulate", c_int32),
		("patch_grid", c_int32), ("img_c", c_int32), ("img_hw", c_int32),
		("tokens_per_image", c_int32), ("lead_rows", c_int32),
		("bias", c_void_p), ("gamma", c_void_p), ("pos", c_void_p),
		("aux", c_void_p), ("ld_aux", c_int64),
		("out2", c_void_p), ("ld_out2", c_int64),
		("a_colsum", c_void_p),
		("workspace", c_void_p), ("workspace_bytes", c_int64),
		("drop_p", c_float * 2), ("drop_seed", ctypes.c_uint64 * 2), ("drop_row0", c_int64),
	]


P, I64, I, FL = c_void_p, c_int64, c_int, c_float

# name -> argtypes (restype is int unless listed in _RESTYPES); order follows include/vitk.h
SIGNATURES = {
	"vitk_strerror": [I],
	"vitk_version": [],
	"vitk_gemm": [POINTER(GemmDesc), P],
	"vitk_gemm_group": [POINTER(GemmDesc), c_int, P],
	"vitk_gemm_uses_bf16_mfma": [POINTER(GemmDesc)],
	"vitk_gemm_last_kernel": [],
	"vitk_gemm_workspace_bytes": [],
	"vitk_gemm_leave_cus_free": [I],
	"vitk_layernorm_fwd": [P, P, P, P, P, P, P, P, I64, I64, FL, I, P],
	"vitk_dropout_epoch_set": [P],
	"vitk_dropout_epoch_advance": [P, P],
	"vitk_layernorm_bwd": [P, P, P, P, P, P, P, P, P, I64, I64, I, P, I64, P],
	"vitk_rmsnorm_fwd": [P, P, P, P, P, P, I64, I64, FL, I, I, P],
	"vitk_rmsnorm_bwd": [P, P, P, P, P, P, P, I64, I64, I, I, P, I64, P],
	"vitk_dyt_fwd": [P, P, P, P, P, I64, I64, I, P],
	"vitk_dyt_bwd": [P, P, P, P, P, P, P, P, I64, I64, I, P],
	"vitk_attention_fwd": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, FL, I, P],
	"vitk_attention_bwd": [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, FL, P, I, P],
	"vitk_attention_bwd_one_pass": [I],
	"vitk_attention_fwd_dropout": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, FL, FL, ctypes.c_uint64, I, P],
	"vitk_attention_bwd_dropout": [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, FL, FL, ctypes.c_uint64, P, I, P],
	"vitk_attention_bwd_noalloc": [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, FL, FL, ctypes.c_uint64, I, P],
	"vitk_scale_by": [P, P, P, I64, I, P],
	"vitk_cast": [P, P, I64, I, I, P],
	"vitk_transpose": [P, P, I64, I64, I64, I64, I, P],
	"vitk_transpose_batched": [P, P, I64, I, P],
	"vitk_dropout": [P, P, I64, FL, ctypes.c_uint64, I, P],
	"vitk_dropout_add": [P, P, P, I64, FL, ctypes.c_uint64, FL, ctypes.c_uint64, I, P],
	"vitk_colsum": [P, P, I64, I64, I64, I, I, P],
	"vitk_gated_act_fwd": [P, P, I64, I64, I64, I, I, P],
	"vitk_gated_act_bwd": [P, P, P, I64, I64, I64, I, I, P],
	"vitk_act_fwd": [P, P, I64, I, I, P],
	"vitk_act_bwd": [P, P, P, I64, I, I, P],
	"vitk_scale_residual": [P, P, P, P, I64, I64, I, P],
	"vitk_layerscale_bwd": [P, P, P, P, P, I64, I64, I, P],
	"vitk_cls_rows": [P, P, P, I64, I64, I64, I, P],
	"vitk_batch_sum": [P, P, I64, I64, I64, I, I, P],
	"vitk_patchify": [P, P, I64, I64, I64, I64, I, P],
	"vitk_unpatchify": [P, P, I64, I64, I64, I64, I, P],
	"vitk_drop_cls_rows": [P, P, I64, I64, I64, I, P],
	"vitk_drop_lead_rows": [P, P, I64, I64, I64, I64, I, P],
	"vitk_add_rows": [P, P, I64, I64, I64, I64, I, P],
	"vitk_mean_pool_fwd": [P, P, I64, I64, I64, I, P],
	"vitk_mean_pool_bwd": [P, P, I64, I64, I64, I, P],
	"vitk_ce_kd_loss": [P, P, P, P, P, I64, I64, FL, FL, FL, I, P],
	"vitk_dual_head_fwd": [P, P, P, P, P, P, P, I64, I64, I64, I64, I, P],
	"vitk_deit_dual_loss": [P, P, P, P, P, P, P, I64, I64, FL, FL, FL, I, P],
	"vitk_adamw": [P, P, P, P, P, P, I64, FL, FL, FL, FL, FL, I64, FL, P],
	"vitk_adamw_device_step": [P, P, P, P, P, P, I64, FL, FL, FL, FL, FL, P, FL, P],
	"vitk_adamw_device_hparams": [P, P, P, P, P, P, I64, P, P, FL, P],
	"vitk_adamw_advance": [P, P, P],
	"vitk_adamw_range": [P, P, P, P, P, P, I64, P, P, FL, I, P],
	"vitk_dwconv_fwd": [P, P, P, I64, I64, I64, I64, I64, I64, I64, I, P],
	"vitk_dwconv_bwd": [P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I, P],
	"vitk_dwconv_fwd_ld": [P, I64, P, P, I64, I64, I64, I64, I64, I64, I64, I, P],
	"vitk_dwconv_bwd_ld": [P, P, I64, P, P, I64, P, I64, I64, I64, I64, I64, I64, I64, I, P],
	"vitk_dwconv_bwd_acc": [P, P, I64, P, P, I64, P, I, I64, I64, I64, I64, I64, I64, I64, I, P],
	"vitk_im2col": [P, P, I64, I64, I64, I64, I64, I64, I64, I64, I, I, P],
	"vitk_col2im": [P, P, I64, I64, I64, I64, I64, I64, I64, I64, I, I, P],
	"vitk_pad_cols": [P, P, I64, I64, I64, I64, I, I, P],
	"vitk_batchnorm_fwd": [P, P, P, P, P, P, P, P, I64, I64, FL, FL, I, I, P],
	"vitk_batchnorm_bwd": [P, P, P, P, P, P, P, P, I64, I64, I, I, P],
	"vitk_batchnorm_workspace_bytes": [I64, I64],
	"vitk_batchnorm_fwd_ws": [P, P, P, P, P, P, P, P, I64, I64, FL, FL, I, I, P, I64, P],
	"vitk_batchnorm_bwd_ws": [P, P, P, P, P, P, P, P, I64, I64, I, I, P, I64, P],
	"vitk_batchnorm_act_fwd": [P, P, P, P, P, P, P, P, I64, I64, FL, FL, I, I, I, P, I64, P],
	"vitk_batchnorm_act_bwd": [P, P, P, P, P, P, P, P, P, I64, I64, I, I, I, P, I64, P],
}
_RESTYPES = {"vitk_strerror": c_char_p, "vitk_gemm_workspace_bytes": c_int64, "vitk_gemm_last_kernel": c_char_p, "vitk_batchnorm_workspace_bytes": c_int64}

_lib = None


def load():
	"""Load libvitk.so once and attach argtypes.  Raises if the library has not been built."""
	global _lib
	if _lib is not None:
		return _lib
	if not os.path.exists(LIB_PATH):
		raise RuntimeError(
			f"{LIB_PATH} is missing: build the HIP kernels first (python -c 'import __graft_entry__ as g; g.build()' "
			"or make -C vit_amd/csrc).  vit_amd has no fallback compute path.")
	lib = ctypes.CDLL(LIB_PATH)
	for name, argtypes in SIGNATURES.items():
		fn = getattr(lib, name)  # AttributeError here = header/library mismatch
		fn.argtypes = argtypes
		fn.restype = _RESTYPES.get(name, c_int)
	_lib = lib
	return lib


def check(rc):
	if rc != 0:
		raise RuntimeError(load().vitk_strerror(rc).decode())


def dtype_code(dt):
	if dt == torch.float32:
		return F32
	if dt == torch.bfloat16:
		return BF16
	raise TypeError(f"vit_amd kernels support float32 and bfloat16, got {dt}")


def ptr(t):
	"""Device pointer of a tensor (None -> NULL)."""
	return None if t is None else t.data_ptr()


def require_gpu(*tensors):
	for t in tensors:
		if t is not None and not t.is_cuda:
			raise RuntimeError("vit_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback; the CPU oracle lives under oracle/ for tests)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
	"""Handle of torch's current HIP stream on the current device.  Asked for on every launch: the two direct bindings cost
	a fraction of a microsecond, torch.cuda.current_stream() (a Stream object per call) about 9 us, which is a third of
	the host time of a launch-bound step."""
	if _raw_stream is not None and _raw_device is not None:
		return _raw_stream(_raw_device())
	return torch.cuda.current_stream().cuda_stream
