// A stand-in for a communication kernel that holds some CUs while the training step runs: `n_wg` workgroups, each taking
// a whole CU's worth of LDS (so nothing that needs LDS shares the CU) and spinning for `ticks` of the 100 MHz wall clock.
// Used by tools/diag/contention.py to measure what the persistent GEMM kernels lose when CUs are taken from them.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void hog_kernel(long long ticks, int* sink) {
	extern __shared__ char lds[];
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) {
	}
	if (ticks < 0) sink[0] = lds[threadIdx.x];
}

extern "C" int hog_launch(int n_wg, long long ticks, int lds_bytes, void* stream) {
	static bool attr = false;
	if (!attr) {
		if (hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
		attr = true;
	}
	hipLaunchKernelGGL(hog_kernel, dim3(n_wg), dim3(256), lds_bytes, (hipStream_t)stream, ticks, (int*)nullptr);
	return hipGetLastError() == hipSuccess ? 0 : -2;
}
