"""vit_amd — MI355X-native ViT/CvT hot path behind the reference's nn.Module interfaces.

`vit_amd.layers.*` mirror the reference's `layers/*` modules (same class names, constructor
arguments, parameter names and return types); every forward/backward runs on hand-written gfx950
kernels in libvitk.so through the C ABI of include/vitk.h.  There is no CPU or PyTorch-kernel
fallback: CPU tensors are rejected.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "layers", "models"]
