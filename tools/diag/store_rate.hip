// Diagnostic: how fast can one CU push a 256x256 bf16 tile (128 KB) to memory, by store shape?
// hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/diag/store_rate.hip -o diag_bin/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(512) void k(char* out, int64_t ld, int tiles_n, int reps, int same, long long* stamps) {
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const long long t0 = wall_clock64();
	for (int r = 0; r < reps; ++r) {
		const int tile = same ? blockIdx.x : r * gridDim.x + blockIdx.x;
		char* base = out + (int64_t)(tile / tiles_n) * 256 * ld + (int64_t)(tile % tiles_n) * 512;
		if (MODE == 0 || MODE == 3) {  // 16 B per lane, rows of 512 B: a wave-instruction = 2 full rows
#pragma unroll
			for (int it = 0; it < 16; ++it) {
				const int v = it * 512 + tid;
				const int row = v >> 5, c = v & 31;
				uint4 val = {(unsigned)v, 1u, 2u, 3u};
				uint4* p = reinterpret_cast<uint4*>(base + row * ld + c * 16);
				typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
				if (MODE == 3) __builtin_nontemporal_store((u32x4){val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4*>(p)); else *p = val;
			}
		} else if (MODE == 1) {  // accumulator layout: 8 B per lane, a wave-instruction = 16 rows x 32 B
			const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
#pragma unroll
			for (int i = 0; i < 8; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					uint2 val = {(unsigned)i, (unsigned)j};
					*reinterpret_cast<uint2*>(base + (wm + 16 * i + (lane & 15)) * ld + (wn + 16 * j + 4 * (lane >> 4)) * 2) = val;
				}
		} else if (MODE == 2) {  // 4 B per lane, a wave-instruction = 256 contiguous bytes
#pragma unroll 8
			for (int it = 0; it < 64; ++it) {
				const int v = it * 512 + tid;
				const int row = v >> 7, c = v & 127;
				*reinterpret_cast<unsigned*>(base + row * ld + c * 4) = (unsigned)v;
			}
		} else if (MODE == 5) {  // per-wave 128x64 sub-tile, 16 B per lane: a wave-instruction = 8 rows x 128 B
			const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
#pragma unroll
			for (int it = 0; it < 16; ++it) {
				const int row = wm + it * 8 + (lane >> 3), c = lane & 7;
				uint4 val = {(unsigned)it, 1u, 2u, 3u};
				*reinterpret_cast<uint4*>(base + row * ld + wn * 2 + c * 16) = val;
			}
		} else if (MODE == 6) {  // per-wave 128x64 sub-tile, 16 B per lane: a wave-instruction = 16 rows x 64 B (lanes row + 16 q)
			const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
#pragma unroll
			for (int it = 0; it < 16; ++it) {
				const int row = wm + (it >> 1) * 16 + (lane & 15), c = (lane >> 4) + 4 * (it & 1);
				uint4 val = {(unsigned)it, 1u, 2u, 3u};
				*reinterpret_cast<uint4*>(base + row * ld + wn * 2 + c * 16) = val;
			}
		} else if (MODE == 4) {  // 16 B per lane, a wave-instruction = 4 rows x 256 B (half rows)
#pragma unroll
			for (int it = 0; it < 16; ++it) {
				const int v = it * 512 + tid;
				const int row = (v >> 4) & 255, c = (v & 15) + 16 * (v >> 12);
				uint4 val = {(unsigned)v, 1u, 2u, 3u};
				*reinterpret_cast<uint4*>(base + row * ld + c * 16) = val;
			}
		}
	}
	const long long t1 = wall_clock64();
	__builtin_amdgcn_s_waitcnt(0);
	asm volatile("s_waitcnt vmcnt(0)");
	const long long t2 = wall_clock64();
	if (tid == 0) {
		stamps[blockIdx.x * 2] = t1 - t0;
		stamps[blockIdx.x * 2 + 1] = t2 - t0;
	}
}

template <int MODE>
void run(const char* name, char* out, int64_t ld, int tiles_n, int blocks, int same, long long* stamps) {
	const int reps = 8;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	float best = 1e9;
	for (int it = 0; it < 3; ++it) {
		hipEventRecord(e0, 0);
		hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, ld, tiles_n, reps, same, stamps);
		hipEventRecord(e1, 0);
		hipDeviceSynchronize();
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	long long h[2 * 256];
	hipMemcpy(h, stamps, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
	double a = 0, b = 0;
	for (int i = 0; i < blocks; ++i) { a += h[2 * i] * 0.01; b += h[2 * i + 1] * 0.01; }
	printf("%-34s blocks %3d %s: kernel %7.1f us, issue %6.2f us/tile, issue+drain %6.2f us/tile -> %5.1f GB/s per CU, %5.2f TB/s chip\n", name, blocks, same ? "L2-resident" : "streaming  ",
		best * 1e3, a / blocks / reps, b / blocks / reps, 131072.0 / (b / blocks / reps) / 1e3, 131072.0 * blocks / (b / blocks / reps) / 1e6);
}

int main() {
	const int tiles_n = 9, tiles_m = 256;  // N = 2304 bf16
	const int64_t ld = tiles_n * 512;
	char* out;
	hipMalloc(&out, (size_t)tiles_m * 256 * ld);
	long long* stamps;
	hipMalloc(&stamps, 8 * 2 * 256);
	for (int blocks : {256, 64}) {
		for (int same : {0, 1}) {
			run<0>("16B/lane rows (2 rows x 512B)", out, ld, tiles_n, blocks, same, stamps);
			run<3>("16B/lane rows, nontemporal", out, ld, tiles_n, blocks, same, stamps);
			run<4>("16B/lane (4 rows x 256B)", out, ld, tiles_n, blocks, same, stamps);
			run<5>("16B/lane per-wave (8 rows x 128B)", out, ld, tiles_n, blocks, same, stamps);
			run<6>("16B/lane per-wave (16 rows x 64B)", out, ld, tiles_n, blocks, same, stamps);
			run<1>("8B/lane fragments (16 rows x 32B)", out, ld, tiles_n, blocks, same, stamps);
			run<2>("4B/lane (256B contiguous)", out, ld, tiles_n, blocks, same, stamps);
		}
	}
	return 0;
}
