// Status strings and version of the vitk C ABI.
#include <hip/hip_runtime.h>

#include "vitk.h"

extern "C" const char* vitk_strerror(int status) {
	switch (status) {
		case VITK_OK: return "ok";
		case VITK_ERR_SHAPE: return "vitk: a dimension is non-positive or outside the supported range";
		case VITK_ERR_DTYPE: return "vitk: unsupported dtype code";
		case VITK_ERR_ALIGN: return "vitk: pointer or leading dimension is not aligned as required";
		case VITK_ERR_NULL: return "vitk: a required pointer is null";
		case VITK_ERR_UNSUPPORTED: return "vitk: unsupported option combination";
		case VITK_ERR_SCRATCH: return "vitk: the problem needs more scratch than the library's fixed buffer holds (use the entry point that takes a caller buffer)";
		default: break;
	}
	if (status <= VITK_ERR_HIP) return hipGetErrorString((hipError_t)(VITK_ERR_HIP - status));
	return "vitk: unknown status";
}

extern "C" int vitk_version(void) { return VITK_VERSION; }

// ---- dropout epoch: one registered device word per device (see DropKey::epoch in vitk_common.h) ------------------------
#include <atomic>
#include <stdint.h>

namespace {
constexpr int MAX_DEVICES = 64;
std::atomic<const uint32_t*> drop_epoch[MAX_DEVICES];

__global__ void epoch_advance_kernel(uint32_t* epoch) {
	if (threadIdx.x == 0) *epoch = *epoch + 1u;
}
}  // namespace

const uint32_t* vitk_dropout_epoch_current(void) {
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return nullptr;
	return drop_epoch[dev].load(std::memory_order_relaxed);
}

extern "C" int vitk_dropout_epoch_set(const uint32_t* epoch) {
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	if (dev < 0 || dev >= MAX_DEVICES) return VITK_ERR_UNSUPPORTED;
	drop_epoch[dev].store(epoch, std::memory_order_relaxed);
	return VITK_OK;
}

extern "C" int vitk_dropout_epoch_advance(uint32_t* epoch, void* stream) {
	if (epoch == nullptr) return VITK_ERR_NULL;
	hipLaunchKernelGGL(epoch_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, epoch);
	hipError_t e = hipGetLastError();
	return e == hipSuccess ? VITK_OK : VITK_ERR_HIP - (int)e;
}
