// Diagnostic: does an LDS-DMA load count in lgkmcnt?  One wave issues a DMA, then waits on lgkmcnt(0) or vmcnt(0).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

__global__ void k(const char* src, long long* out) {
	__shared__ __attribute__((aligned(16))) char smem[8192];
	const int lane = threadIdx.x;
	for (int rep = 0; rep < 4; ++rep) {
		const char* p = src + (size_t)rep * 65536 + lane * 16;
		__syncthreads();
		long long t0 = __builtin_amdgcn_s_memtime();
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_global_load_lds((gbl_void_p)p, (lds_void_p)smem, 16, 0, 0);
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		long long t1 = __builtin_amdgcn_s_memtime();
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		long long t2 = __builtin_amdgcn_s_memtime();
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		if (lane == 0) {
			out[rep * 2] = t1 - t0;
			out[rep * 2 + 1] = t2 - t0;
		}
	}
}
int main() {
	char* src;
	long long* out;
	hipMalloc(&src, 1 << 20);
	hipMemset(src, 1, 1 << 20);
	hipMalloc(&out, 64);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
	long long h[8];
	hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
	for (int r = 0; r < 4; ++r) printf("rep %d: after lgkmcnt(0) %lld cycles, after vmcnt(0) %lld cycles\n", r, h[2 * r], h[2 * r + 1]);
	return 0;
}
