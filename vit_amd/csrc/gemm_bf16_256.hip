// Large-tile bf16 MFMA GEMM: 256x256x64 tile per 512-thread workgroup (8 waves as 2 (M) x 4 (N), 128x64 per wave =
// 8x4 tiles of v_mfma_f32_16x16x32_bf16, 128 accumulator registers), operands brought HBM -> LDS by the LDS-DMA
// instruction global_load_lds_dwordx4 (no VGPR staging, no ds_write), two LDS stages of 64 KiB, one workgroup per CU.
//
// An LDS-DMA wave-instruction writes 64 lanes x 16 bytes LINEARLY at a wave-uniform LDS base, but each lane's GLOBAL
// source address is free.  The bank-conflict swizzle of the LDS image is therefore applied to the source address
// (lane l of a piece fetches the chunk that belongs at physical position l) and, with the same XOR, to the fragment
// reads.  K-major operand image: [256 rows][64 k] in 128-byte rows, chunk ^= row & 7 (ds_read_b128).
// MN-major operand image: [64 k][256 cols] in 512-byte rows, 32-byte unit ^= (k & 3) | ((k >> 3) & 1) << 2
// (ds_read_b64_tr_b16, the hardware-transposing read).
//
// The epilogue goes through LDS in two halves of 128 rows so that C and the epilogue operands move in whole lines.
#include <cstdlib>
#include <type_traits>

#include "gemm_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
constexpr int TILE_BYTES = 256 * 64 * 2;  // 32 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int CT_LD = 260;                // floats per row of the epilogue image (256 + 4 pad)
constexpr int CB_LD = 264;                // bf16 per row of the bf16 epilogue image (256 + 8 pad)
constexpr int LDS_BYTES = 256 * CB_LD * 2;  // 135,168 B >= float image of 128 rows (133,120 B) >= 2 stages (131,072 B)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

__device__ __forceinline__ int fk(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }
__device__ __forceinline__ int off_k(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int off_mn(int krow, int chunk) { return krow * 512 + (((((chunk >> 1) ^ fk(krow)) << 1) | (chunk & 1)) << 4); }

template <int LAYOUT>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int base, int s, int lane) {
	if (LAYOUT == VITK_K_MAJOR) {
		return *reinterpret_cast<const bf16x8*>(tile + off_k(base + (lane & 15), 4 * s + (lane >> 4)));
	} else {
		const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
		const int krow = 32 * s + 8 * g + q;
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

__device__ __forceinline__ uint32_t lds_addr(const char* p) {
	return (uint32_t)(uintptr_t)(lds_void_p)(void*)p;
}

__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
	union {
		s16x4 h[2];
		bf16x8 v;
	} u;
	u.h[0] = lo;
	u.h[1] = hi;
	return u.v;
}

// read_frag<MN-major> as two inline-asm transposing reads (issued, NOT waited for: the caller retires them)
__device__ __forceinline__ void read_frag_mn_asm(uint32_t tile_addr, int base, int s, int lane, s16x4& lo, s16x4& hi) {
	const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
	const int krow = 32 * s + 8 * g + q;
	const int col = base + 4 * p;
	const uint32_t addr = tile_addr + off_mn(krow, col >> 3) + ((p & 1) << 3);
	asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
	asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi) : "v"(addr));  // rows +4: same swizzle key
}

// LDS-DMA of one 256 x 64 operand tile: 32 pieces of 1 KiB, 4 per wave.  Every source address is a wave-uniform
// 64-bit base (scalar registers) plus a 32-bit per-lane byte offset that is recomputed from the lane id on each call
// (the empty asm keeps the compiler from hoisting 16 loop-invariant address registers into the MFMA loop, where the
// accumulators leave no room for them).
template <int LAYOUT>
__device__ __forceinline__ void stage_piece(char* tile, const bf16_t* __restrict__ base, int64_t ld, int64_t mn0, int64_t mn_total, int k0, int piece, int lane) {
	asm volatile("" : "+v"(lane));
	const int ldi = (int)ld;
	{
		const char* src;
		if (LAYOUT == VITK_K_MAJOR) {
			const int r = lane >> 3;  // == row & 7
			const int chunk = (lane & 7) ^ r;
			int64_t rowbase = mn0 + 8 * piece;  // rows past the end re-read the last row (their products are never stored)
			if (rowbase > mn_total - 1) rowbase = mn_total - 1;
			const int rlim = (int)(mn_total - 1 - rowbase);
			const int rr = r < rlim ? r : rlim;
			const uint32_t voff = (uint32_t)(rr * ldi + 8 * chunk) * 2u;
			src = reinterpret_cast<const char*>(base + rowbase * ld + k0) + voff;
		} else {
			const int hi = lane >> 5, pc = lane & 31;
			const int krow_u = 2 * piece;
			const int chunk = ((((pc >> 1) ^ fk(krow_u + hi)) << 1) | (pc & 1));
			const int clim = (int)(mn_total - 8 - mn0);  // >= 0: MN-major extents are multiples of 8
			const int coff = 8 * chunk < clim ? 8 * chunk : clim;
			const uint32_t voff = (uint32_t)(hi * ldi + coff) * 2u;
			src = reinterpret_cast<const char*>(base + (int64_t)(k0 + krow_u) * ld + mn0) + voff;
		}
		__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(tile + piece * 1024), 16, 0, 0);
	}
}

template <int LAYOUT>
__device__ __forceinline__ void stage_tile(char* tile, const bf16_t* __restrict__ base, int64_t ld, int64_t mn0, int64_t mn_total, int k0, int wave, int lane) {
#pragma unroll
	for (int j = 0; j < 4; ++j) stage_piece<LAYOUT>(tile, base, ld, mn0, mn_total, k0, 4 * wave + j, lane);
}

// Compile-time loop: the body sees its index as a constant, so accumulator arrays indexed by it can never become
// runtime-indexed (which would move them to scratch memory) even when the optimizer declines to unroll a large body.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		static_for<I + 1, N>(f);
	}
}

// One K step (64) of the 256x256 tile for one wave: 2 x (4 B fragments, 8 A fragments, 32 MFMAs).  The fragment reads
// are software-pipelined two A fragments ahead of the MFMAs that use them, and sched_group_barrier pins that order
// (left alone, the compiler issues each read right before its first use and the MFMA pipe idles for the LDS latency).
// Measured on the ViT-B/16 shapes: +5..15 % with a K-major A operand; with an MN-major A operand (two transposing
// reads per fragment, weight-gradient GEMMs) the compiler's own order is 3 % faster and is kept.
struct NoDma {
	__device__ __forceinline__ void operator()(int) const {}
};

// `dma(j)`, j = 0..7: issues the j-th LDS-DMA request of the next K step; they go out one per two MFMA groups.
template <int ALAYOUT, int BLAYOUT, bool SPREAD_DMA = false, typename DMA = NoDma>
__device__ __forceinline__ void mma_stage(const char* sa, const char* sb, f32x4 (&acc)[8][4], int wm, int wn, int lane, DMA&& dma = NoDma{}) {
	if constexpr (ALAYOUT == VITK_K_MAJOR) {
		constexpr int RB = BLAYOUT == VITK_K_MAJOR ? 1 : 2;  // LDS reads per B fragment
		bf16x8 wf[2][4], xf[3];
#pragma unroll
		for (int j = 0; j < 4; ++j) wf[0][j] = read_frag<BLAYOUT>(sb, wn + 16 * j, 0, lane);
		xf[0] = read_frag<ALAYOUT>(sa, wm, 0, lane);
		xf[1] = read_frag<ALAYOUT>(sa, wm + 16, 0, lane);
		static_for<0, 16>([&](auto gc) {
			constexpr int g = decltype(gc)::value, s = g >> 3, i = g & 7;
			if constexpr (SPREAD_DMA && (g & 1) == 0) dma(g >> 1);
			if constexpr (g + 2 < 16) xf[(g + 2) % 3] = read_frag<ALAYOUT>(sa, wm + 16 * ((g + 2) & 7), (g + 2) >> 3, lane);
			if constexpr (g == 4) {
#pragma unroll
				for (int j = 0; j < 4; ++j) wf[1][j] = read_frag<BLAYOUT>(sb, wn + 16 * j, 1, lane);
			}
#pragma unroll
			for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], xf[g % 3], acc[i][j], 0, 0, 0);
		});
		// the order above, spelled out for the instruction scheduler: 0x100 = LDS reads, 0x008 = MFMAs
		__builtin_amdgcn_sched_group_barrier(0x100, 4 * RB + 2, 0);
		static_for<0, 16>([&](auto gc) {
			constexpr int g = decltype(gc)::value;
			constexpr int reads = (g + 2 < 16 ? 1 : 0) + (g == 4 ? 4 * RB : 0);
			if constexpr (SPREAD_DMA && (g & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
			if constexpr (reads > 0) __builtin_amdgcn_sched_group_barrier(0x100, reads, 0);
			__builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
		});
	} else if constexpr (BLAYOUT == VITK_MN_MAJOR) {
		// Both operands MN-major (weight gradients).  The transposing reads are issued as inline asm: for the builtin the
		// compiler puts an s_waitcnt vmcnt(0) in front of the first read after an LDS-DMA request (it cannot tell that the
		// request goes to the OTHER stage), which would make every K step wait for the next one's operands.  The asm
		// reads carry no such ordering; their results are retired by an explicit lgkmcnt(0) that names all fragment
		// registers, so no MFMA can be scheduled above it.
		if constexpr (SPREAD_DMA) {
#pragma unroll
			for (int j = 0; j < 8; ++j) dma(j);
		}
		const uint32_t abase = lds_addr(sa), bbase = lds_addr(sb);
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			s16x4 xh[8][2], wh[4][2];
#pragma unroll
			for (int j = 0; j < 4; ++j) read_frag_mn_asm(bbase, wn + 16 * j, s, lane, wh[j][0], wh[j][1]);
#pragma unroll
			for (int i = 0; i < 8; ++i) read_frag_mn_asm(abase, wm + 16 * i, s, lane, xh[i][0], xh[i][1]);
			asm volatile("s_waitcnt lgkmcnt(0)"
						 : "+v"(wh[0][0]), "+v"(wh[0][1]), "+v"(wh[1][0]), "+v"(wh[1][1]), "+v"(wh[2][0]), "+v"(wh[2][1]), "+v"(wh[3][0]), "+v"(wh[3][1]), "+v"(xh[0][0]),
						   "+v"(xh[0][1]), "+v"(xh[1][0]), "+v"(xh[1][1]), "+v"(xh[2][0]), "+v"(xh[2][1]), "+v"(xh[3][0]), "+v"(xh[3][1]), "+v"(xh[4][0]), "+v"(xh[4][1]),
						   "+v"(xh[5][0]), "+v"(xh[5][1]), "+v"(xh[6][0]), "+v"(xh[6][1]), "+v"(xh[7][0]), "+v"(xh[7][1]));
#pragma unroll
			for (int i = 0; i < 8; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(wh[j][0], wh[j][1]), join8(xh[i][0], xh[i][1]), acc[i][j], 0, 0, 0);
		}
	} else {
		if constexpr (SPREAD_DMA) {
#pragma unroll
			for (int j = 0; j < 8; ++j) dma(j);
		}
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			bf16x8 xf[8], wf[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) wf[j] = read_frag<BLAYOUT>(sb, wn + 16 * j, s, lane);
#pragma unroll
			for (int i = 0; i < 8; ++i) xf[i] = read_frag<ALAYOUT>(sa, wm + 16 * i, s, lane);
#pragma unroll
			for (int i = 0; i < 8; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
		}
	}
}

// Activation in the bf16-output epilogues: GELU known at compile time takes the transcendental-free form.
template <int ACT>
__device__ __forceinline__ float act_fwd(float u, int act_rt) {
	if constexpr (ACT == VITK_ACT_GELU) return gelu_bf16(u);
	else return act_apply(u, act_rt);
}
template <int ACT>
__device__ __forceinline__ float act_bwd(float u, int act_rt) {
	if constexpr (ACT == VITK_ACT_GELU) return gelu_deriv_bf16(u);
	else return act_deriv(u, act_rt);
}

struct BigArgs {
	GemmDev g;
	int tiles_m, tiles_n, splits, k_per_split;
	int vec4, vec8;  // epilogue operands / outputs may be accessed 4-wide (8 / 16 bytes) resp. 8-wide (16 bytes bf16)
	float* a_colsum;  // MN-major A: sum over k of A(m, k) is added here (bias gradient), or null
	float* cs_partials;  // with partials: [split][tile column][row half][M] partial column sums for split_reduce_kernel instead of atomics
	float* partials;  // K-split partial tiles [split][tile][wave][i][j][lane][4] (accumulator order), or null: float atomics into C
#ifdef VITK_GEMM_STAMPS
	long long* stamps;  // diagnostic builds only: [workgroup][round][8] timestamps of the persistent kernel
	int stagger;        // diagnostic builds only: start delay step in 10 ns ticks
#endif
};

#ifdef VITK_GEMM_STAMPS
long long* g_stamps = nullptr;
int g_stagger = 0;
#define STAMP(slot) \
	if (a.stamps != nullptr && tid == 0 && round < 16) a.stamps[((int64_t)blockIdx.x * 16 + round) * 8 + (slot)] = wall_clock64()
#else
#define STAMP(slot)
#endif

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

template <int ALAYOUT, int BLAYOUT, typename TOUT, int EPI, int ACT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_256_kernel(BigArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	gemm_dev_enter(a.g);
	const GemmDev& g = a.g;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;

	// 1-D grid over (K slice, tile), K slice outermost: workgroups that share an XCD (equal blockIdx % 8) get a
	// contiguous run of that order, i.e. neighbouring tiles of the SAME K slice, whose operand panels they then share
	// through the XCD's L2.
	const int ntile = a.tiles_m * a.tiles_n;
	const int nwg = ntile * a.splits;
	int wg = blockIdx.x;
	{
		const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
		wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
	}
	const int split = wg / ntile;
	wg -= split * ntile;
	constexpr int GM = 4;
	const int per_group = GM * a.tiles_n;
	const int group = wg / per_group;
	const int first_m = group * GM;
	const int gsz = min(a.tiles_m - first_m, GM);
	const int tm = first_m + (wg % per_group) % gsz;
	const int tn = (wg % per_group) / gsz;
	int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

	const int k_begin = split * a.k_per_split;
	const int k_end = min((int)g.K, k_begin + a.k_per_split);
	const int nt = (k_end - k_begin) / BK;

	const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
	const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);

	f32x4 acc[8][4];
#pragma unroll
	for (int i = 0; i < 8; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#ifdef VITK_GEMM_STAMPS
	const int round = 0;
#endif
	STAMP(0);
	stage_tile<ALAYOUT>(smem, A, g.lda, m0, g.M, k_begin, wave, lane);
	stage_tile<BLAYOUT>(smem + TILE_BYTES, B, g.ldb, n0, g.N, k_begin, wave, lane);
	__syncthreads();  // with LDS-DMA in flight hipcc drains vmcnt(0) ahead of the barrier
	STAMP(1);

	// Column sums of an MN-major A (the bias gradient next to a weight gradient): the A tile of a K step is in LDS
	// anyway; the tiles_n workgroups that share it take turns (K step kt belongs to the one with tn == kt % tiles_n),
	// each of its 512 threads adds up 32 of the 64 rows of one column.
	float csum = 0.f;
	const bool do_colsum = ALAYOUT == VITK_MN_MAJOR && a.a_colsum != nullptr;
	for (int t = 0; t < nt; ++t) {
		const char* sa = smem + (t & 1) * STAGE_BYTES;
		const char* sb = sa + TILE_BYTES;
		// (before the next stage's LDS-DMA is issued: the compiler orders plain LDS loads behind every LDS-DMA in flight,
		// which would make this K step wait for the next one's operands)
		if constexpr (ALAYOUT == VITK_MN_MAJOR) {
			if (do_colsum && (k_begin / BK + t) % a.tiles_n == tn) {
				const int c = tid & 255, kr0 = (tid >> 8) * 32;
#pragma unroll 8
				for (int kr = kr0; kr < kr0 + 32; ++kr) csum += (float)*reinterpret_cast<const bf16_t*>(sa + off_mn(kr, c >> 3) + (c & 7) * 2);
			}
		}
		if (t + 1 < nt) {
			char* da = smem + ((t + 1) & 1) * STAGE_BYTES;
			const int k0 = k_begin + (t + 1) * BK;
			stage_tile<ALAYOUT>(da, A, g.lda, m0, g.M, k0, wave, lane);
			stage_tile<BLAYOUT>(da + TILE_BYTES, B, g.ldb, n0, g.N, k0, wave, lane);
		}
		mma_stage<ALAYOUT, BLAYOUT>(sa, sb, acc, wm, wn, lane);
		__syncthreads();  // next stage landed (vmcnt(0) drained by the fence) and everyone is done with this one
	}

	if constexpr (ALAYOUT == VITK_MN_MAJOR) {
		if (do_colsum && m0 + (tid & 255) < g.M) {
			// (float atomics of ~170 workgroup halves on one address run one after the other at the memory side)
			if (a.cs_partials != nullptr) a.cs_partials[(((int64_t)split * a.tiles_n + tn) * 2 + (tid >> 8)) * g.M + m0 + (tid & 255)] = csum;
			else atomicAdd(a.a_colsum + m0 + (tid & 255), csum);
		}
	}
	STAMP(2);
#ifdef VITK_GEMM_STAMPS
	if (a.stagger == -1) m0 = n0 = 0;
#endif
	if constexpr (std::is_same<TOUT, bf16_t>::value) {
		// bf16 output: the fused epilogue runs in the accumulator layout (each lane owns 4 consecutive columns of one
		// row: float4 bias, 8-byte residual / pre-activation operand), the result is rounded to bf16 and the whole
		// 256x256 tile goes through ONE LDS image, then leaves in 16-byte stores, 512 contiguous bytes per row.
		// acc[i][j][r]: m = wm + 16 i + (lane & 15), n = wn + 16 j + 4 (lane >> 4) + r
		bf16_t* img = reinterpret_cast<bf16_t*>(smem);
		float bias4[4][4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int64_t nb = n0 + wn + 16 * j + 4 * (lane >> 4);
#pragma unroll
			for (int r = 0; r < 4; ++r) bias4[j][r] = (g.bias != nullptr && nb + r < g.N) ? g.bias[nb + r] : 0.f;
		}
		const bf16_t* aux = reinterpret_cast<const bf16_t*>(g.aux);
		bf16_t* out2 = reinterpret_cast<bf16_t*>(g.out2);
		static_for<0, 8>([&](auto ic) {
			static_for<0, 4>([&](auto jc) {
				constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
				const int ml = wm + 16 * i + (lane & 15);
				const int nl = wn + 16 * j + 4 * (lane >> 4);
				const int64_t m = m0 + ml, nb = n0 + nl;
				const bool in = m < g.M && nb < g.N;
				const bool full = a.vec4 && nb + 3 < g.N;
				float v[4];
#pragma unroll
				for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bias4[j][r];
				float keep[4] = {1.f, 1.f, 1.f, 1.f};  // fused dropout
				if (EPI != VITK_EPI_NONE && g.drop[0].thr16) drop_keep_n<4>(g, m, nb, keep);
				if (EPI == VITK_EPI_RESIDUAL || EPI == VITK_EPI_ACTGRAD) {
#pragma unroll
					for (int r = 0; r < 4; ++r) v[r] *= keep[r];
				}
				if (EPI == VITK_EPI_ACT) {
					if (out2 != nullptr) {
						// the pre-activation leaves straight from the accumulator layout (8 bytes per lane); the activation
						// sees it as stored (rounded to bf16), which is what the backward will read
						bf16x4 pre;
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							pre[r] = (bf16_t)v[r];
							v[r] = (float)pre[r];
						}
						if (in) {
							if (full) {
								*reinterpret_cast<bf16x4*>(out2 + m * g.ld_out2 + nb) = pre;
							} else {
#pragma unroll
								for (int r = 0; r < 4; ++r)
									if (nb + r < g.N) out2[m * g.ld_out2 + nb + r] = pre[r];
							}
						}
					}
#pragma unroll
					for (int r = 0; r < 4; ++r) v[r] = keep[r] * act_fwd<ACT>(v[r], g.act);
				} else if (EPI == VITK_EPI_RESIDUAL || EPI == VITK_EPI_ACTGRAD) {
					float x4[4] = {0.f, 0.f, 0.f, 0.f};
					if (in) {
						if (full) {
							load_vec<bf16_t, 4>(aux + m * g.ld_aux + nb, x4);
						} else {
#pragma unroll
							for (int r = 0; r < 4; ++r)
								if (nb + r < g.N) x4[r] = (float)aux[m * g.ld_aux + nb + r];
						}
					}
					if (EPI == VITK_EPI_RESIDUAL) {
#pragma unroll
						for (int r = 0; r < 4; ++r) v[r] = x4[r] + ((g.gamma != nullptr && nb + r < g.N) ? g.gamma[nb + r] : 1.f) * v[r];
					} else {
#pragma unroll
						for (int r = 0; r < 4; ++r) v[r] *= act_bwd<ACT>(x4[r], g.act);
					}
				}
				bf16x4 pk;
#pragma unroll
				for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
				*reinterpret_cast<bf16x4*>(img + ml * CB_LD + nl) = pk;
			});
		});
		__syncthreads();
		STAMP(3);
		bf16_t* outp = reinterpret_cast<bf16_t*>(g.C);
		for (int it = 0; it < (BM * BN / 8) / NTHREADS; ++it) {
			const int v = it * NTHREADS + tid;
			const int row = v >> 5, c8 = v & 31;
			const int64_t m = m0 + row, nb = n0 + 8 * c8;
			if (m >= g.M || nb >= g.N) continue;
			const uint4 val = *reinterpret_cast<const uint4*>(img + row * CB_LD + 8 * c8);
			if (a.vec8 && nb + 7 < g.N) {
				*reinterpret_cast<uint4*>(outp + m * g.ldc + nb) = val;
			} else {
				const bf16_t* e = reinterpret_cast<const bf16_t*>(&val);
#pragma unroll
				for (int r = 0; r < 8; ++r)
					if (nb + r < g.N) outp[m * g.ldc + nb + r] = e[r];
			}
		}
		STAMP(4);
	} else {
		if (a.splits > 1 && a.partials != nullptr) {
			// K-split partial tile: straight from the accumulators in their own order, every wave-instruction one
			// contiguous KiB; split_reduce_kernel adds the slices up and undoes the order
			float* dst = a.partials + ((int64_t)split * ntile + (int64_t)tm * a.tiles_n + tn) * (BM * BN) + (int64_t)wave * (32 * 256) + lane * 4;
#pragma unroll
			for (int i = 0; i < 8; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dst + (i * 4 + j) * 256) = acc[i][j];
			return;
		}
		// float32 output (weight gradients, split-K partial sums): through a float LDS image, 128 rows at a time
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
				// split-K partial sums: float atomics, one dword per lane on consecutive addresses (256 contiguous bytes
				// per wave-instruction, the shape the memory-side atomic units take at full rate)
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
}

// Epilogue of one wave's 128x64 accumulator tile through a wave-PRIVATE float LDS image, 32 rows at a time, with no
// workgroup barrier: the wave writes 8 fragments (accumulator layout: 16 rows x 16 bytes per wave-instruction), reads
// them back as rows (each lane 8 consecutive columns) and does the fused math there, so that the epilogue operands
// arrive and the results leave in 16-byte accesses, 8 rows x 128 contiguous bytes per wave-instruction.  (In the
// accumulator layout itself a wave-instruction touches 16 rows x 32 bytes, and the CU's memory pipeline takes about as
// long per instruction for those 512 bytes as for 1 KiB in whole lines.)
constexpr int EW_LD = 68;                    // floats per image row: 64 + 4 pad (conflict-free for both access shapes)
constexpr int EW_BYTES = 32 * EW_LD * 4;     // 8,704 B per wave
constexpr int LDSP_BYTES = STAGE_BYTES + 8 * EW_BYTES;  // stage 0 stays free for the next tile's first K step

// Per-tile epilogue operands a lane keeps in registers: bias / LayerScale of its 8 columns and the first 32 rows of the
// residual / pre-activation operand.  Loaded BEFORE the next tile's LDS-DMA is issued: the memory counter retires in
// order, so anything requested after the DMA could only be consumed once the DMA has landed.
struct EpiRegs {
	float bias8[8], gamma8[8];
	bf16x8 aux0[4];
};

template <int EPI>
__device__ __forceinline__ void epilogue_prefetch(const BigArgs& a, EpiRegs& er, bool full, int64_t m0w, int64_t n0w, int lane) {
	const GemmDev& g = a.g;
	const int c8 = lane & 7, r8 = lane >> 3;
	const int64_t nb = n0w + 8 * c8;
#pragma unroll
	for (int e = 0; e < 8; ++e) {
		er.bias8[e] = 0.f;
		er.gamma8[e] = 1.f;
	}
	if (!full) return;
	if (g.bias != nullptr) {
		load_vec<float, 4>(g.bias + nb, *reinterpret_cast<float(*)[4]>(&er.bias8[0]));
		load_vec<float, 4>(g.bias + nb + 4, *reinterpret_cast<float(*)[4]>(&er.bias8[4]));
	}
	if (EPI == VITK_EPI_RESIDUAL && g.gamma != nullptr) {
		load_vec<float, 4>(g.gamma + nb, *reinterpret_cast<float(*)[4]>(&er.gamma8[0]));
		load_vec<float, 4>(g.gamma + nb + 4, *reinterpret_cast<float(*)[4]>(&er.gamma8[4]));
	}
	if (EPI == VITK_EPI_RESIDUAL || EPI == VITK_EPI_ACTGRAD) {
		const bf16_t* aux = reinterpret_cast<const bf16_t*>(g.aux);
#pragma unroll
		for (int it = 0; it < 4; ++it) {
			const int64_t m = m0w + 8 * it + r8;
			er.aux0[it] = *reinterpret_cast<const bf16x8*>(aux + m * g.ld_aux + nb);
		}
	}
}

// `full`: wave-uniform, the wave's 128 rows x 64 columns are all inside M x N and every operand may be accessed 16
// bytes wide; then the whole epilogue is straight-line code.
template <int EPI, int ACT>
__device__ __forceinline__ void epilogue_wave_private(const BigArgs& a, const f32x4 (&acc)[8][4], const EpiRegs& er, bool full, float* img, int64_t m0w, int64_t n0w, int lane) {
	const GemmDev& g = a.g;
	const int c8 = lane & 7, r8 = lane >> 3;
	const int64_t nb = n0w + 8 * c8;
	const bf16_t* aux = reinterpret_cast<const bf16_t*>(g.aux);
	bf16_t* out2 = reinterpret_cast<bf16_t*>(g.out2);
	bf16_t* outp = reinterpret_cast<bf16_t*>(g.C);
	constexpr bool HAS_AUX = EPI == VITK_EPI_RESIDUAL || EPI == VITK_EPI_ACTGRAD;
	auto write_quarter = [&](auto qc) {
		constexpr int qq = decltype(qc)::value;
#pragma unroll
		for (int ii = 0; ii < 2; ++ii)
#pragma unroll
			for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(img + (16 * ii + (lane & 15)) * EW_LD + 16 * j + 4 * (lane >> 4)) = acc[2 * qq + ii][j];
	};
	// The same wave reads what it wrote: LDS operations of one wave complete in order, so no barrier anywhere.
	if (!full) {
		// ragged edge tile or unaligned operands: one element at a time through the generic epilogue
		static_for<0, 4>([&](auto qc) {
			constexpr int qq = decltype(qc)::value;
			write_quarter(qc);
#pragma unroll 1
			for (int idx = lane; idx < 32 * 64; idx += 64) {
				const int row = idx >> 6, col = idx & 63;
				const int64_t m = m0w + 32 * qq + row, n = n0w + col;
				if (m < g.M && n < g.N) epi_store1<bf16_t, bf16_t, EPI, ACT>(g, m, n, img[row * EW_LD + col]);
			}
		});
		return;
	}
	// Operand rows (residual / pre-activation) of the NEXT quarter are requested at the top of each quarter, as inline
	// asm with a wave-uniform row base in scalar registers and one 32-bit lane offset, and retired by a COUNTED wait
	// that leaves this quarter's requests and the previous quarter's stores in flight.  (Written as plain loads, the
	// compiler sinks them to their uses and waits vmcnt(0) there, i.e. for every store issued before them.)  The
	// "memory" clobbers pin the stores between the asm statements, which the count relies on.
	const bool drop_on = g.drop[0].thr16 != 0;
	u32x4 auxv[2][4];
	const uint32_t aux_voff = (uint32_t)(r8 * (int)g.ld_aux + 8 * c8) * 2u;
	if (HAS_AUX) {
#pragma unroll
		for (int it = 0; it < 4; ++it) auxv[0][it] = __builtin_bit_cast(u32x4, er.aux0[it]);
	}
	static_for<0, 4>([&](auto qc) {
		constexpr int qq = decltype(qc)::value;
		if (HAS_AUX && qq < 3) {
#pragma unroll
			for (int it = 0; it < 4; ++it) {
				const bf16_t* rowbase = aux + (m0w + 32 * (qq + 1) + 8 * it) * g.ld_aux + n0w;  // wave-uniform
				asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(auxv[(qq + 1) & 1][it]) : "v"(aux_voff), "s"(rowbase) : "memory");
			}
		}
		if (HAS_AUX && qq > 0) {
			// issued after the rows needed now: 4 stores of the previous quarter (+ 4 requests just above)
			u32x4(&cur)[4] = auxv[qq & 1];
			if constexpr (qq < 3) asm volatile("s_waitcnt vmcnt(8)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
			else asm volatile("s_waitcnt vmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3])::"memory");
		}
		write_quarter(qc);
#pragma unroll
		for (int it = 0; it < 4; ++it) {
			const int row = 8 * it + r8;
			const int64_t m = m0w + 32 * qq + row;
			const f32x4 lo = *reinterpret_cast<const f32x4*>(img + row * EW_LD + 8 * c8);
			const f32x4 hi = *reinterpret_cast<const f32x4*>(img + row * EW_LD + 8 * c8 + 4);
			float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
			for (int e = 0; e < 8; ++e) v[e] += er.bias8[e];
			float keep[8];
			if (EPI != VITK_EPI_NONE && drop_on) {  // fused dropout (wave-uniform branch)
				drop_keep_n<8>(g, m, nb, keep);
				if (EPI != VITK_EPI_ACT) {
#pragma unroll
					for (int e = 0; e < 8; ++e) v[e] *= keep[e];
				}
			}
			if (EPI == VITK_EPI_ACT) {
				if (out2 != nullptr) {
					bf16x8 pre;
#pragma unroll
					for (int e = 0; e < 8; ++e) {
						pre[e] = (bf16_t)v[e];
						v[e] = (float)pre[e];  // the activation sees the pre-activation as stored
					}
					*reinterpret_cast<bf16x8*>(out2 + m * g.ld_out2 + nb) = pre;
				}
				if constexpr (ACT == VITK_ACT_GELU) {
					gelu_bf16_n<8>(v);
				} else {
#pragma unroll
					for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e], g.act);
				}
				if (drop_on) {
#pragma unroll
					for (int e = 0; e < 8; ++e) v[e] *= keep[e];
				}
			} else if (EPI == VITK_EPI_RESIDUAL) {
#pragma unroll
				for (int e = 0; e < 8; ++e) v[e] = (float)__builtin_bit_cast(bf16x8, auxv[qq & 1][it])[e] + er.gamma8[e] * v[e];
			} else if (EPI == VITK_EPI_ACTGRAD) {
				float u[8];
#pragma unroll
				for (int e = 0; e < 8; ++e) u[e] = (float)__builtin_bit_cast(bf16x8, auxv[qq & 1][it])[e];
				if constexpr (ACT == VITK_ACT_GELU) {
					gelu_deriv_bf16_n<8>(v, u);
				} else {
#pragma unroll
					for (int e = 0; e < 8; ++e) v[e] *= act_deriv(u[e], g.act);
				}
			}
			bf16x8 pk;
#pragma unroll
			for (int e = 0; e < 8; ++e) pk[e] = (bf16_t)v[e];
			*reinterpret_cast<bf16x8*>(outp + m * g.ldc + nb) = pk;
		}
	});
}

// Persistent form for bf16 outputs with more tiles than CUs: one workgroup per CU walks the tile list; after the last
// K step of a tile it first issues the LDS-DMA of the NEXT tile's first K step (into stage 0) and only then runs the
// epilogue (through stage 1 and the spare LDS), so output stores, epilogue operand loads and the next tile's operand
// fetch overlap and no workgroup is launched or drained between tiles.
template <int ALAYOUT, int BLAYOUT, int EPI, int ACT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_256p_kernel(BigArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	gemm_dev_enter(a.g);
	const GemmDev& g = a.g;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
	const int ntiles = a.tiles_m * a.tiles_n;
	const int nblk = gridDim.x;  // a multiple of 8
	const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
	constexpr int GM = 4;
	const int per_group = GM * a.tiles_n;

	// tile of this workgroup in a round: consecutive tiles (in grouped order) go to workgroups that share an XCD
	auto tile_of = [&](int round, int64_t& m0, int64_t& n0) -> bool {
		const int base = round * nblk;
		const int n = min(nblk, ntiles - base);
		if (n <= 0) return false;
		const int q = n >> 3, r = n & 7;
		if (idx >= q + (xcd < r ? 1 : 0)) return false;
		const int wg = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
		const int group = wg / per_group;
		const int first_m = group * GM;
		const int gsz = min(a.tiles_m - first_m, GM);
		const int tm = first_m + (wg % per_group) % gsz;
		const int tn = (wg % per_group) / gsz;
		m0 = (int64_t)tm * BM;
		n0 = (int64_t)tn * BN;
		return true;
	};

	const int nt = (int)(g.K / BK);
	const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
	const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);

	// LDS: [stage buffer 0: 64 KiB][4 KiB][stage buffer 1: 64 KiB].  The K steps of consecutive tiles form ONE pipeline:
	// during the last K step of a tile the first K step of the next tile is already being fetched into the other
	// buffer; the epilogue image takes the buffer that step just vacated plus the 4 KiB in the middle.
	constexpr int BUF1 = STAGE_BYTES + (8 * EW_BYTES - STAGE_BYTES);
	int64_t m0 = 0, n0 = 0;
	bool have = tile_of(0, m0, n0);
#ifdef VITK_GEMM_STAMPS
	if (a.stagger > 0) {
		const long long until = wall_clock64() + (long long)(idx & 3) * a.stagger;
		while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
	}
#endif
	int par = 0;  // buffer of the current K step
	if (have) {
		stage_tile<ALAYOUT>(smem, A, g.lda, m0, g.M, 0, wave, lane);
		stage_tile<BLAYOUT>(smem + TILE_BYTES, B, g.ldb, n0, g.N, 0, wave, lane);
	}
	for (int round = 0; have; ++round) {
		STAMP(0);
		f32x4 acc[8][4];
#pragma unroll
		for (int i = 0; i < 8; ++i)
#pragma unroll
			for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
		int64_t nm0 = m0, nn0 = n0;  // after the last tile: re-fetch this tile's first K step (nobody reads it)
		bool nxt = false;
		for (int t = 0; t < nt; ++t) {
			// this step's operands have landed (the barrier's fence drains the DMA) and every wave is done with the
			// other buffer: the previous K step, or the previous tile's epilogue image
			__syncthreads();
			const char* sa = smem + (par ? BUF1 : 0);
			const char* sb = sa + TILE_BYTES;
			char* da = smem + (par ? 0 : BUF1);
			const bool last = t + 1 == nt;
			if (last) nxt = tile_of(round + 1, nm0, nn0);
			const int64_t sm0 = last ? nm0 : m0, sn0 = last ? nn0 : n0;
			const int sk0 = last ? 0 : (t + 1) * BK;
			// Issue of the 8 LDS-DMA requests of the next K step: spread between the MFMA groups when B is MN-major
			// (measured +5..9 % there), all up front when both operands are K-major (spreading costs registers and is
			// neutral to slightly negative there).
			constexpr bool SPREAD = BLAYOUT == VITK_MN_MAJOR;
			if constexpr (!SPREAD) {
				stage_tile<ALAYOUT>(da, A, g.lda, sm0, g.M, sk0, wave, lane);
				stage_tile<BLAYOUT>(da + TILE_BYTES, B, g.ldb, sn0, g.N, sk0, wave, lane);
			}
			mma_stage<ALAYOUT, BLAYOUT, SPREAD>(sa, sb, acc, wm, wn, lane, [&](int j) {
				if (j & 1) stage_piece<BLAYOUT>(da + TILE_BYTES, B, g.ldb, sn0, g.N, sk0, 4 * wave + (j >> 1), lane);
				else stage_piece<ALAYOUT>(da, A, g.lda, sm0, g.M, sk0, 4 * wave + (j >> 1), lane);
			});
			par ^= 1;
		}
		STAMP(2);
		// now `par` names the buffer the next tile's first K step is landing in; the other one was read last
		const bool full = a.vec8 && n0 + wn + 64 <= g.N && m0 + wm + 128 <= g.M;
		EpiRegs er;
		epilogue_prefetch<EPI>(a, er, full, m0 + wm, n0 + wn, lane);
		__syncthreads();  // every wave is done with the last K step's buffer (and the requests above have arrived)
		STAMP(3);
		float* img = reinterpret_cast<float*>(smem + (par ? 0 : STAGE_BYTES) + wave * EW_BYTES);
		epilogue_wave_private<EPI, ACT>(a, acc, er, full, img, m0 + wm, n0 + wn, lane);
		STAMP(4);
		m0 = nm0;
		n0 = nn0;
		have = nxt;
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last (unused) DMA must not outlive the workgroup's LDS
}

// C (+)= sum over the K slices of the partial tiles written above.  One thread per accumulator quad (4 consecutive
// columns of one row): 16-byte loads at unit stride per slice, one 16-byte store.
__global__ __launch_bounds__(256) void split_reduce_kernel(const float* __restrict__ partials, float* __restrict__ C, int64_t M, int64_t N, int64_t ldc, int tiles_n,
	int ntile, int splits, int accumulate, int vec4, const float* __restrict__ cs_partials, float* a_colsum) {
	if ((int)blockIdx.y == ntile) {  // the extra grid row: a_colsum[m] += sum of the partial column sums
		if (cs_partials == nullptr) return;
		// 16 columns per block, the parts dealt to 16 thread groups, so that a block has all its loads in flight at once
		const int parts = splits * tiles_n * 2;
		__shared__ float red[16][17];
		const int grp = threadIdx.x >> 4, mi = threadIdx.x & 15;
		for (int64_t mb = (int64_t)blockIdx.x * 16; mb < M; mb += (int64_t)gridDim.x * 16) {
			const int64_t m = mb + mi;
			float sum = 0.f;
			if (m < M) {
#pragma unroll 4
				for (int p = grp; p < parts; p += 16) sum += cs_partials[(int64_t)p * M + m];
			}
			red[grp][mi] = sum;
			__syncthreads();
			if (grp == 0 && m < M) {
				float tot = 0.f;
#pragma unroll
				for (int r = 0; r < 16; ++r) tot += red[r][mi];
				a_colsum[m] += tot;
			}
			__syncthreads();
		}
		return;
	}
	const int tile = blockIdx.y;
	const int q = blockIdx.x * 256 + threadIdx.x;  // quad index inside the tile: ((wave * 8 + i) * 4 + j) * 64 + lane
	const int lane = q & 63, j = (q >> 6) & 3, i = (q >> 8) & 7, wave = q >> 11;
	const int64_t m = (int64_t)(tile / tiles_n) * BM + (wave >> 2) * 128 + 16 * i + (lane & 15);
	const int64_t n = (int64_t)(tile % tiles_n) * BN + (wave & 3) * 64 + 16 * j + 4 * (lane >> 4);
	if (m >= M || n >= N) return;
	const float* src = partials + (int64_t)tile * (BM * BN) + (int64_t)q * 4;
	// eight slices requested before the first is used: written as load-add-load-add the loop is a chain of memory
	// round trips (14 us for 28 slices of a 768 x 768 gradient; 6 us this way)
	const int64_t step = (int64_t)ntile * (BM * BN);
	f32x4 part[8];
#pragma unroll
	for (int u = 0; u < 8; ++u) part[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
	for (int s0 = 0; s0 < splits; s0 += 8) {
		f32x4 t[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) t[u] = s0 + u < splits ? *reinterpret_cast<const f32x4*>(src + (int64_t)(s0 + u) * step) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int u = 0; u < 8; ++u) part[u] += t[u];
	}
	f32x4 v = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
	float* c = C + m * ldc + n;
	if (vec4 && n + 3 < N) {
		if (accumulate) v += *reinterpret_cast<const f32x4*>(c);
		*reinterpret_cast<f32x4*>(c) = v;
	} else {
#pragma unroll
		for (int r = 0; r < 4; ++r)
			if (n + r < N) c[r] = accumulate ? c[r] + v[r] : v[r];
	}
}

template <int AL, int BL, typename TOUT, int EPI, int ACT>
int launch1(const BigArgs& a, dim3 grid, hipStream_t st) {
	static bool attr = false;
	if (!attr) {
		hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_256_kernel<AL, BL, TOUT, EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr = true;
	}
	hipLaunchKernelGGL((gemm_bf16_256_kernel<AL, BL, TOUT, EPI, ACT>), grid, dim3(NTHREADS), LDS_BYTES, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

int persistent_blocks() {
	static int n = 0;
	if (n == 0) {
		int dev = 0, cus = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
		n = cus & ~7;
#ifdef VITK_GEMM_STAMPS
		if (const char* e = std::getenv("VITK_PBLOCKS")) n = atoi(e);
#endif
	}
	return n;
}

template <int AL, int BL, int EPI, int ACT>
int launchp(const BigArgs& a, hipStream_t st) {
	static bool attr = false;
	if (!attr) {
		hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_256p_kernel<AL, BL, EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSP_BYTES);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr = true;
	}
	hipLaunchKernelGGL((gemm_bf16_256p_kernel<AL, BL, EPI, ACT>), dim3((unsigned)persistent_blocks()), dim3(NTHREADS), LDSP_BYTES, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

// bf16 outputs: persistent kernel when there is more than one tile per CU, else one workgroup per tile
template <int AL, int BL, int EPI, int ACT>
int launch_bf16(const BigArgs& a, dim3 grid, hipStream_t st) {
	static const bool no_persist = std::getenv("VITK_NO_PERSISTENT") != nullptr;
	if (!no_persist && a.splits == 1 && (int64_t)a.tiles_m * a.tiles_n > persistent_blocks()) return launchp<AL, BL, EPI, ACT>(a, st);
	return launch1<AL, BL, bf16_t, EPI, ACT>(a, grid, st);
}

// Variants instantiated (what the ViT / CvT training step uses); any other combination returns VITK_ERR_UNSUPPORTED
// and runs on the 128x128 kernel with its run-time epilogue.
//   (K, K)   forward:  plain (bf16 / f32), bias + activation (+ pre-activation), bias + LayerScale + residual;
//            dgrad against a transposed weight copy: plain, activation derivative
//   (K, MN)  dgrad:    plain, activation derivative
//   (MN, MN) wgrad:    f32 (plain or split-K atomics)
int launch_variant(const BigArgs& a, const vitk_gemm_desc& d, dim3 grid, hipStream_t st) {
	const bool f32out = d.out_dtype == VITK_F32;
	constexpr int K = VITK_K_MAJOR, MN = VITK_MN_MAJOR;
	if (f32out) {
		if (d.epilogue != VITK_EPI_NONE) return VITK_ERR_UNSUPPORTED;
		if (d.a_layout == K && d.b_layout == K) return launch1<K, K, float, VITK_EPI_NONE, -1>(a, grid, st);
		if (d.a_layout == MN && d.b_layout == MN) return launch1<MN, MN, float, VITK_EPI_NONE, -1>(a, grid, st);
		return VITK_ERR_UNSUPPORTED;
	}
	if (d.accumulate) return VITK_ERR_UNSUPPORTED;
	if (d.a_layout == K && d.b_layout == K) {
		switch (d.epilogue) {
			case VITK_EPI_NONE: return launch_bf16<K, K, VITK_EPI_NONE, -1>(a, grid, st);
			case VITK_EPI_ACT:
				return d.act == VITK_ACT_GELU ? launch_bf16<K, K, VITK_EPI_ACT, VITK_ACT_GELU>(a, grid, st) : launch_bf16<K, K, VITK_EPI_ACT, -1>(a, grid, st);
			case VITK_EPI_RESIDUAL: return launch_bf16<K, K, VITK_EPI_RESIDUAL, -1>(a, grid, st);
			case VITK_EPI_ACTGRAD:  // input gradient against a transposed weight copy
				return d.act == VITK_ACT_GELU ? launch_bf16<K, K, VITK_EPI_ACTGRAD, VITK_ACT_GELU>(a, grid, st) : launch_bf16<K, K, VITK_EPI_ACTGRAD, -1>(a, grid, st);
			default: return VITK_ERR_UNSUPPORTED;
		}
	}
	if (d.a_layout == K && d.b_layout == MN) {
		switch (d.epilogue) {
			case VITK_EPI_NONE: return launch_bf16<K, MN, VITK_EPI_NONE, -1>(a, grid, st);
			case VITK_EPI_ACTGRAD:
				return d.act == VITK_ACT_GELU ? launch_bf16<K, MN, VITK_EPI_ACTGRAD, VITK_ACT_GELU>(a, grid, st)
											  : launch_bf16<K, MN, VITK_EPI_ACTGRAD, -1>(a, grid, st);
			default: return VITK_ERR_UNSUPPORTED;
		}
	}
	return VITK_ERR_UNSUPPORTED;
}

}  // namespace

bool vitk_gemm_bf16_256_ok(const vitk_gemm_desc& d) {
	if (d.patch_grid > 0 || d.tokens_per_image > 0 || d.K % BK != 0) return false;
	if (d.M < 256 || d.N < 256) return false;
	if (d.a_layout == VITK_MN_MAJOR && d.M < 8) return false;
	return true;
}

// A persistent launch whose last round of tiles would fill only part of the chip (N = 768 at 197 row tiles: 591 tiles =
// 2.31 rounds of 256) is cut along M: the rows that make whole rounds stay on the persistent kernel, the remaining rows
// go to the 128x128 kernel, whose four times smaller tiles spread that remainder over all CUs (measured on the
// ViT-B/16 shapes: +2..5 % for K >= 2304, where the per-tile work is long; -3 % at K = 768, so short-K problems are
// left alone).
static int split_off_partial_round(const vitk_gemm_desc& d, hipStream_t st, bool* handled) {
	*handled = false;
	static const bool off = std::getenv("VITK_NO_TAIL_SPLIT") != nullptr;
	if (off || d.out_dtype != VITK_BF16 || d.a_layout != VITK_K_MAJOR || d.accumulate || d.K < 1536) return VITK_OK;
	if (d.patch_grid > 0 || d.tokens_per_image > 0) return VITK_OK;  // rows are not plain rows of A / C there
	const int64_t P = persistent_blocks();
	const int64_t tiles_m = vitk_cdiv(d.M, BM), tiles_n = vitk_cdiv(d.N, BN);
	const int64_t tiles = tiles_m * tiles_n;
	if (tiles <= P) return VITK_OK;
	const int64_t rounds = tiles / P, rem = tiles % P;
	if (rem == 0 || rem * 10 > P * 6) return VITK_OK;
	const int64_t tm1 = (rounds * P) / tiles_n;  // row tiles that fit into the whole rounds
	const int64_t M1 = tm1 * BM;
	if (tm1 <= 0 || M1 >= d.M) return VITK_OK;
	vitk_gemm_desc d1 = d, d2 = d;
	d1.M = M1;
	d2.M = d.M - M1;
	const size_t e = sizeof(bf16_t);
	d2.A = reinterpret_cast<const char*>(d.A) + (size_t)M1 * d.lda * e;
	d2.C = reinterpret_cast<char*>(d.C) + (size_t)M1 * d.ldc * e;
	if (d.aux != nullptr) d2.aux = reinterpret_cast<const char*>(d.aux) + (size_t)M1 * d.ld_aux * e;
	if (d.out2 != nullptr) d2.out2 = reinterpret_cast<char*>(d.out2) + (size_t)M1 * d.ld_out2 * e;
	d2.drop_row0 = d.drop_row0 + M1;  // the second block draws the mask of its rows in the whole tensor
	*handled = true;
	bool unused = false;
	int rc = vitk_gemm_bf16_256(d1, st, &unused);
	if (rc != VITK_OK) return rc;
	return vitk_gemm_bf16_128(d2, st);
}

int vitk_gemm_bf16_256(const vitk_gemm_desc& d, hipStream_t st, bool* colsum_done) {
	{
		bool handled = false;
		const int rc = split_off_partial_round(d, st, &handled);
		if (handled) {
			vitk_last_gemm_kernel = "gemm_bf16_256_kernel";  // (the bulk of the split launch)
			return rc;
		}
	}
	vitk_last_gemm_kernel = "gemm_bf16_256_kernel";
	BigArgs a{};
	a.g = make_gemm_dev(d);
	a.a_colsum = d.a_colsum;
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
		static const char* force = std::getenv("VITK_SPLITK");  // tuning / diagnosis
		if (force != nullptr) a.splits = std::max(1, atoi(force));
	}
	const int64_t kt_per = vitk_cdiv(ktiles, a.splits);
	a.k_per_split = (int)(kt_per * BK);
	a.splits = (int)vitk_cdiv(ktiles, kt_per);
	auto al16 = [](const void* p) { return p == nullptr || vitk_aligned(p, 16); };
	a.vec4 = (d.N % 4 == 0) && (d.ldc % 4 == 0) && (d.ld_aux % 4 == 0) && (d.ld_out2 % 4 == 0) && al16(d.C) && al16(d.bias) && al16(d.gamma) && al16(d.pos) &&
		al16(d.aux) && al16(d.out2);
	a.vec8 = a.vec4 && (d.N % 8 == 0) && (d.ldc % 8 == 0) && (d.ld_out2 % 8 == 0);
#ifdef VITK_GEMM_STAMPS
	a.stamps = g_stamps;
	a.stagger = g_stagger;
#endif
	const int64_t need = (int64_t)a.splits * tiles * BM * BN * (int64_t)sizeof(float);
	const bool cs_fused = d.a_colsum != nullptr && d.a_layout == VITK_MN_MAJOR && d.b_layout == VITK_MN_MAJOR && d.out_dtype == VITK_F32;
	const int64_t cs_need = cs_fused ? (int64_t)a.splits * a.tiles_n * 2 * d.M * (int64_t)sizeof(float) : 0;
	if (a.splits > 1 && d.workspace != nullptr && d.workspace_bytes >= need + cs_need && vitk_aligned(d.workspace, 16)) {
		a.partials = reinterpret_cast<float*>(d.workspace);
		if (cs_fused) a.cs_partials = a.partials + need / (int64_t)sizeof(float);
	}
	if (a.splits > 1 && a.partials == nullptr && !d.accumulate) {
		hipError_t e = vitk_zero_f32_2d(reinterpret_cast<float*>(d.C), d.M, d.N, d.ldc, st);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	}
	int rc = VITK_ERR_UNSUPPORTED;
	static const bool p8w = std::getenv("VITK_NO_P8W") == nullptr;  // A-B switch
	if (p8w && a.partials != nullptr && d.a_layout == VITK_MN_MAJOR && d.b_layout == VITK_MN_MAJOR && d.out_dtype == VITK_F32 && (d.a_colsum == nullptr || a.cs_partials != nullptr)) {
		WgradArgs w{};
		w.g = a.g;
		w.tiles_m = a.tiles_m;
		w.tiles_n = a.tiles_n;
		w.splits = a.splits;
		w.k_per_split = a.k_per_split;
		w.cs_partials = a.cs_partials;
		w.partials = a.partials;
		if (vitk_gemm_bf16_p8w_ok(w)) rc = vitk_gemm_bf16_p8w(w, st);
	}
	if (rc == VITK_ERR_UNSUPPORTED) rc = launch_variant(a, d, dim3((unsigned)(tiles * a.splits)), st);
	if (rc == VITK_OK && a.partials != nullptr) {
		hipLaunchKernelGGL(split_reduce_kernel, dim3(BM * BN / 4 / 256, (unsigned)tiles + 1), dim3(256), 0, st, a.partials, reinterpret_cast<float*>(d.C), d.M, d.N, d.ldc, a.tiles_n,
			(int)tiles, a.splits, d.accumulate, a.vec4, a.cs_partials, d.a_colsum);
		VITK_CHECK_LAUNCH();
	}
	// only the float32-output kernel with MN-major A (weight gradients) carries the column sums
	if (rc == VITK_OK && d.a_colsum != nullptr && d.a_layout == VITK_MN_MAJOR && d.b_layout == VITK_MN_MAJOR && d.out_dtype == VITK_F32) *colsum_done = true;
	return rc;
}

// Split-K launches at most one workgroup per CU (plus rounding), each with one 256x256 float32 partial tile.
extern "C" int64_t vitk_gemm_workspace_bytes(void) {
	return (int64_t)(persistent_blocks() + 64) * BM * BN * (int64_t)sizeof(float);
}
