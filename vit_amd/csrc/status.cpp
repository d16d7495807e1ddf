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
