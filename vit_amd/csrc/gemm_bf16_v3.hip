// Large-tile bf16 MFMA GEMM, pipelined variant: 256x256 output tile per 512-thread workgroup (8 waves as 2 (M) x 4 (N),
// 128x64 per wave = 8x4 tiles of v_mfma_f32_16x16x32_bf16, 128 accumulator registers), one workgroup per CU.
//
// K is walked in steps of 32 through a RING OF FOUR LDS stages (A 16 KiB + B 16 KiB each) filled by LDS-DMA
// (global_load_lds_dwordx4).  In iteration s the workgroup
//   1. waits, with a COUNTED s_waitcnt vmcnt(4), until its own pieces of stage s+1 have landed (stage s+2 stays in
//      flight) and passes one raw s_barrier: stage s+1 is now visible to all waves, and everybody is done with stage s-1;
//   2. issues the DMA of stage s+3 into the slot of stage s-1;
//   3. reads the fragments of step s+1 from LDS into a second register set, under
//   4. the 32 MFMAs of step s, whose fragments were read one iteration earlier.
// So the DMA queue never drains inside the loop and the MFMAs of a step never wait for LDS after the barrier.
//
// An LDS-DMA wave-instruction writes 64 lanes x 16 bytes LINEARLY at a wave-uniform LDS base, but each lane's GLOBAL
// source address is free.  The bank-conflict swizzle of the LDS image is therefore applied to the source address
// and, with the same XOR, to the fragment reads.  K-major stage: [256 rows][32 k] in 64-byte rows,
// chunk ^= 3 * ((row >> 2) & 1) (ds_read_b128).  MN-major stage: [32 k][256 cols] in 512-byte rows, 32-byte unit ^=
// (k & 3) | ((k >> 3) & 1) << 2 (ds_read_b64_tr_b16, the hardware-transposing read).
//
// The epilogue goes through LDS in two halves of 128 rows so that C and the epilogue operands move in whole lines.
#include <cstdlib>

#include "gemm_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 32, NTHREADS = 512, NSTAGE = 4;
constexpr int TILE_BYTES = 256 * 32 * 2;  // 16 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int CT_LD = 260;                // floats per row of the epilogue image (256 + 4 pad)
constexpr int LDS_BYTES = 128 * CT_LD * 4;  // 133,120 B (> 4 stages = 131,072 B)

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

__device__ __forceinline__ int fk(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }
__device__ __forceinline__ int off_k(int row, int chunk) { return row * 64 + ((chunk ^ (3 * ((row >> 2) & 1))) << 4); }
__device__ __forceinline__ int off_mn(int krow, int chunk) { return krow * 512 + (((((chunk >> 1) ^ fk(krow)) << 1) | (chunk & 1)) << 4); }

template <int LAYOUT>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int base, int lane) {
	if (LAYOUT == VITK_K_MAJOR) {
		return *reinterpret_cast<const bf16x8*>(tile + off_k(base + (lane & 15), lane >> 4));
	} else {
		const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
		const int krow = 8 * g + q;
		const int col = base + 4 * p;
		const int off = off_mn(krow, col >> 3) + ((p & 1) << 3);
		typedef s16x4 __attribute__((address_space(3))) * lds_ptr_t;
		s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + off));
		s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + off + 4 * 512));  // rows +4: same swizzle key
		union {
			s16x4 h[2];
			bf16x8 v;
		} u;
		u.h[0] = lo;
		u.h[1] = hi;
		return u.v;
	}
}

// LDS-DMA of one operand's 256 x 32 stage: 16 pieces of 1 KiB, 2 per wave.
template <int LAYOUT>
__device__ __forceinline__ void stage_tile(char* tile, const bf16_t* __restrict__ base, int64_t ld, int64_t mn0, int64_t mn_total, int k0, int wave, int lane) {
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		const int piece = 2 * wave + j;
		const bf16_t* src;
		if (LAYOUT == VITK_K_MAJOR) {
			const int rl = lane >> 2;  // row within the piece: 16 rows of 64 bytes
			int64_t row = mn0 + 16 * piece + rl;
			if (row >= mn_total) row = mn_total - 1;
			const int chunk = (lane & 3) ^ (3 * ((rl >> 2) & 1));
			src = base + row * ld + k0 + 8 * chunk;
		} else {
			const int krow = 2 * piece + (lane >> 5);
			const int pc = lane & 31;
			const int chunk = ((((pc >> 1) ^ fk(krow)) << 1) | (pc & 1));
			int64_t col = mn0 + 8 * chunk;
			if (col > mn_total - 8) col = mn_total - 8;
			src = base + (int64_t)(k0 + krow) * ld + col;
		}
		__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(tile + piece * 1024), 16, 0, 0);
	}
}

struct BigArgs {
	GemmDev g;
	int tiles_m, tiles_n, splits, k_per_split;
	int vec4;
};

template <typename TOUT>
__device__ __forceinline__ void store4(const BigArgs& a, int64_t m, int64_t nb, f32x4 v) {
	const GemmDev& g = a.g;
	if (a.vec4 && nb + 3 < g.N) {
		const float f[4] = {v[0], v[1], v[2], v[3]};
		epi_store4<bf16_t, TOUT>(g, m, nb, f);
	} else {
		if (nb + 0 < g.N) epi_store1<bf16_t, TOUT>(g, m, nb + 0, v[0]);
		if (nb + 1 < g.N) epi_store1<bf16_t, TOUT>(g, m, nb + 1, v[1]);
		if (nb + 2 < g.N) epi_store1<bf16_t, TOUT>(g, m, nb + 2, v[2]);
		if (nb + 3 < g.N) epi_store1<bf16_t, TOUT>(g, m, nb + 3, v[3]);
	}
}

template <int ALAYOUT, int BLAYOUT, typename TOUT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_v3_kernel(BigArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const GemmDev& g = a.g;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;

	const int nwg = a.tiles_m * a.tiles_n;
	int wg = blockIdx.x;
	{
		const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
		wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
	}
	constexpr int GM = 4;
	const int per_group = GM * a.tiles_n;
	const int group = wg / per_group;
	const int first_m = group * GM;
	const int gsz = min(a.tiles_m - first_m, GM);
	const int tm = first_m + (wg % per_group) % gsz;
	const int tn = (wg % per_group) / gsz;
	const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

	const int k_begin = blockIdx.y * a.k_per_split;
	const int k_end = min((int)g.K, k_begin + a.k_per_split);
	const int nt = (k_end - k_begin) / BK;

	const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
	const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);

	f32x4 acc[8][4];
#pragma unroll
	for (int i = 0; i < 8; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

	// ---- pipelined main loop (see the header) ----
	auto issue = [&](int step) {  // DMA of K-step `step` (0-based within this workgroup's K range) into its ring slot
		char* st = smem + (step & (NSTAGE - 1)) * STAGE_BYTES;
		const int k0 = k_begin + step * BK;
		stage_tile<ALAYOUT>(st, A, g.lda, m0, g.M, k0, wave, lane);
		stage_tile<BLAYOUT>(st + TILE_BYTES, B, g.ldb, n0, g.N, k0, wave, lane);
	};
	auto load_frags = [&](int step, bf16x8(&xf)[8], bf16x8(&wf)[4]) {
		const char* sa = smem + (step & (NSTAGE - 1)) * STAGE_BYTES;
		const char* sb = sa + TILE_BYTES;
#pragma unroll
		for (int j = 0; j < 4; ++j) wf[j] = read_frag<BLAYOUT>(sb, wn + 16 * j, lane);
#pragma unroll
		for (int i = 0; i < 8; ++i) xf[i] = read_frag<ALAYOUT>(sa, wm + 16 * i, lane);
	};
	auto mma = [&](const bf16x8(&xf)[8], const bf16x8(&wf)[4]) {
#pragma unroll
		for (int i = 0; i < 8; ++i)
#pragma unroll
			for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
	};
	// One pipeline step: fragments of step s are in (xc, wc); (xn, wn_) receive those of step s + 1.
	// Every wave issues 4 DMA pieces per stage, so "all but my 4 youngest" = stage s+1 has landed when s+2 was issued.
	auto step_fn = [&](int s, const bf16x8(&xc)[8], const bf16x8(&wc)[4], bf16x8(&xn)[8], bf16x8(&wn_)[4]) {
		__builtin_amdgcn_sched_barrier(0);
		if (s + 2 < nt)
			asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
		else
			asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		__builtin_amdgcn_sched_barrier(0);
		if (s + 3 < nt) issue(s + 3);
		if (s + 1 < nt) load_frags(s + 1, xn, wn_);
		mma(xc, wc);
	};

	const int nt32 = nt;  // number of 32-deep K-steps (nt is computed with BK = 32)
	(void)nt32;
	issue(0);
	if (nt > 1) issue(1);
	if (nt > 2) issue(2);
	bf16x8 xa[8], wa[4], xb[8], wb[4];
	{
		if (nt > 2)
			asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
		else if (nt > 1)
			asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
		else
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		__builtin_amdgcn_sched_barrier(0);
		load_frags(0, xa, wa);
	}
#pragma unroll 1
	for (int s = 0; s < nt; s += 2) {
		step_fn(s, xa, wa, xb, wb);
		if (s + 1 < nt) step_fn(s + 1, xb, wb, xa, wa);
	}
	__syncthreads();  // all MFMAs done, LDS free for the epilogue image

	// epilogue through LDS, 128 rows at a time
	float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
	for (int ph = 0; ph < 2; ++ph) {
		if ((wave >> 2) == ph) {
#pragma unroll
			for (int i = 0; i < 8; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j)
					*reinterpret_cast<f32x4*>(ct + (16 * i + (lane & 15)) * CT_LD + wn + 16 * j + 4 * (lane >> 4)) = acc[i][j];
		}
		__syncthreads();
		if (a.splits > 1) {
			// split-K partial sums: float atomics, one dword per lane on consecutive addresses (256 contiguous bytes per
			// wave-instruction, the shape the memory-side atomic units take at full rate)
			float* cbase = reinterpret_cast<float*>(g.C);
			for (int it = 0; it < (128 * BN) / NTHREADS; ++it) {
				const int v = it * NTHREADS + tid;
				const int row = v >> 8, col = v & 255;
				const int64_t m = m0 + ph * 128 + row, n = n0 + col;
				if (m < g.M && n < g.N) atomicAdd(cbase + m * g.ldc + n, ct[row * CT_LD + col]);
			}
		} else {
			for (int it = 0; it < (128 * BN / 4) / NTHREADS; ++it) {
				const int v = it * NTHREADS + tid;
				const int row = v >> 6, c4 = v & 63;
				const int64_t m = m0 + ph * 128 + row, nb = n0 + 4 * c4;
				if (m < g.M && nb < g.N) store4<TOUT>(a, m, nb, *reinterpret_cast<const f32x4*>(ct + row * CT_LD + 4 * c4));
			}
		}
		__syncthreads();
	}
}

template <int AL, int BL>
int launch(const BigArgs& a, int out_dtype, dim3 grid, hipStream_t st) {
	static bool attr_set[2] = {false, false};
	const int oi = out_dtype == VITK_BF16 ? 0 : 1;
	if (!attr_set[oi]) {
		hipError_t e = oi == 0 ? hipFuncSetAttribute((const void*)gemm_bf16_v3_kernel<AL, BL, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES)
							   : hipFuncSetAttribute((const void*)gemm_bf16_v3_kernel<AL, BL, float>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr_set[oi] = true;
	}
	if (out_dtype == VITK_BF16)
		hipLaunchKernelGGL((gemm_bf16_v3_kernel<AL, BL, bf16_t>), grid, dim3(NTHREADS), LDS_BYTES, st, a);
	else
		hipLaunchKernelGGL((gemm_bf16_v3_kernel<AL, BL, float>), grid, dim3(NTHREADS), LDS_BYTES, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

}  // namespace

bool vitk_gemm_bf16_v3_ok(const vitk_gemm_desc& d) {
	if (d.patch_grid > 0 || d.K % BK != 0) return false;
	if (d.M < 256 || d.N < 256) return false;
	if (d.a_layout == VITK_MN_MAJOR && d.M < 8) return false;
	return true;
}

int vitk_gemm_bf16_v3(const vitk_gemm_desc& d, hipStream_t st) {
	BigArgs a{};
	a.g = make_gemm_dev(d);
	a.tiles_m = (int)vitk_cdiv(d.M, BM);
	a.tiles_n = (int)vitk_cdiv(d.N, BN);
	const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
	const int64_t ktiles = d.K / BK;
	a.splits = 1;
	// split-K (weight gradients: few output tiles, very long K): one workgroup per CU, as few K slices as fill the
	// chip once, because every extra slice adds a full tile of float atomics
	if (d.out_dtype == VITK_F32 && d.epilogue == VITK_EPI_NONE && d.bias == nullptr && d.tokens_per_image == 0 && tiles <= 128 && ktiles >= 16) {
		int64_t s = std::min<int64_t>(256 / tiles, ktiles / 8);
		a.splits = (int)std::max<int64_t>(1, std::min<int64_t>(s, 64));
	}
	const int64_t kt_per = vitk_cdiv(ktiles, a.splits);
	a.k_per_split = (int)(kt_per * BK);
	a.splits = (int)vitk_cdiv(ktiles, kt_per);
	auto al16 = [](const void* p) { return p == nullptr || vitk_aligned(p, 16); };
	a.vec4 = (d.N % 4 == 0) && (d.ldc % 4 == 0) && (d.ld_aux % 4 == 0) && (d.ld_out2 % 4 == 0) && al16(d.C) && al16(d.bias) && al16(d.gamma) && al16(d.pos) &&
		al16(d.aux) && al16(d.out2);
	if (a.splits > 1 && !d.accumulate) {
		hipError_t e = hipMemset2DAsync(d.C, (size_t)d.ldc * sizeof(float), 0, (size_t)d.N * sizeof(float), (size_t)d.M, st);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	}
	dim3 grid((unsigned)tiles, (unsigned)a.splits);
	if (d.a_layout == VITK_K_MAJOR && d.b_layout == VITK_K_MAJOR) return launch<VITK_K_MAJOR, VITK_K_MAJOR>(a, d.out_dtype, grid, st);
	if (d.a_layout == VITK_K_MAJOR && d.b_layout == VITK_MN_MAJOR) return launch<VITK_K_MAJOR, VITK_MN_MAJOR>(a, d.out_dtype, grid, st);
	if (d.a_layout == VITK_MN_MAJOR && d.b_layout == VITK_MN_MAJOR) return launch<VITK_MN_MAJOR, VITK_MN_MAJOR>(a, d.out_dtype, grid, st);
	return launch<VITK_MN_MAJOR, VITK_K_MAJOR>(a, d.out_dtype, grid, st);
}
