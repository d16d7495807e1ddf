// Diagnostic: per-CU throughput of the two global->LDS paths on L2-resident data (one 512-thread workgroup per CU):
//   mode 0: LDS-DMA (global_load_lds_dwordx4), mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128, mode 2: half/half
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/diag/load_rate.hip -o diag_bin/load_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const char* src, int iters, long long* out, float* sink) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	// each workgroup streams its own 128 KiB window (64 KiB per "K step", two alternating), rows of 128 B at stride 1536 B
	const char* base = src + (size_t)(blockIdx.x % 64) * (256 * 1536);
	const long long t0 = wall_clock64();
	float acc = 0.f;
	for (int it = 0; it < iters; ++it) {
		const int k0 = (it & 7) * 128;  // walk along the row like a K loop
		char* buf = smem + (it & 1) * 65536;
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int piece = 8 * wave + j;  // 64 pieces of 1 KiB = 8 rows x 128 B
			const char* p = base + (size_t)(8 * (piece & 31) + (lane >> 3)) * 1536 + k0 + (piece >> 5) * 768 + (lane & 7) * 16;
			if (MODE == 0 || (MODE == 2 && (j & 1) == 0)) {
				__builtin_amdgcn_global_load_lds((gbl_void_p)p, (lds_void_p)(buf + piece * 1024), 16, 0, 0);
			} else {
				const uint4 v = *reinterpret_cast<const uint4*>(p);
				*reinterpret_cast<uint4*>(buf + piece * 1024 + lane * 16) = v;
			}
		}
		__syncthreads();
		acc += *reinterpret_cast<float*>(smem + (it & 1) * 65536 + tid * 4);
	}
	const long long t1 = wall_clock64();
	if (tid == 0) out[blockIdx.x] = t1 - t0;
	if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
void run(const char* name, const char* src, int blocks, long long* out, float* sink) {
	const int iters = 2000;
	hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
	for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 131072, 0, src, iters, out, sink);
	hipDeviceSynchronize();
	long long h[256];
	hipMemcpy(h, out, 8 * blocks, hipMemcpyDeviceToHost);
	double t = 0;
	for (int i = 0; i < blocks; ++i) t += h[i] * 10e-9;
	t /= blocks;
	printf("%-44s blocks %3d: %6.1f GB/s per CU  (%5.2f TB/s chip), %.2f us per 64 KiB\n", name, blocks, 65536.0 * iters / t / 1e9, 65536.0 * iters * blocks / t / 1e12, t / iters * 1e6);
}

int main() {
	char* src;
	hipMalloc(&src, 64 * 256 * 1536 + 4096);
	hipMemset(src, 1, 64 * 256 * 1536 + 4096);
	long long* out;
	float* sink;
	hipMalloc(&out, 8 * 256);
	hipMalloc(&sink, 4);
	for (int blocks : {256, 32}) {
		run<0>("LDS-DMA dwordx4", src, blocks, out, sink);
		run<1>("global_load_dwordx4 + ds_write_b128", src, blocks, out, sink);
		run<2>("half LDS-DMA, half through registers", src, blocks, out, sink);
	}
	return 0;
}
