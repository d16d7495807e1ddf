// Persistent bf16 MFMA GEMM for the large ViT shapes.
//
//   one 512-thread workgroup per CU walks its share of the 256x256 output tiles; each wave owns a 128x64 sub-tile
//   (8x4 tiles of v_mfma_f32_16x16x32_bf16, 128 accumulator registers);
//   operands stream HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) through two stages of one 64-deep K-tile each
//   (A 32 KiB + B 32 KiB per stage): the DMA of K-tile t+1 is issued before the MFMAs of K-tile t;
//   the stream of K-tiles is continuous ACROSS output tiles: while a tile's epilogue runs, the next tile's first
//   stage is already in flight, and the epilogue's global stores drain under the next tile's MFMAs;
//   the epilogue is applied straight from the accumulator registers (bias, activation + pre-activation, residual +
//   LayerScale, activation derivative: each a compile-time variant), 4 consecutive columns per lane.
//
// LDS images (the XOR swizzles are applied to the DMA SOURCE address, because an LDS-DMA piece lands linearly, and to
// the fragment reads):
//   K-major operand stage:  [256 rows][64 k], 128-byte rows, chunk ^= row & 7                 (ds_read_b128)
//   MN-major operand stage: [64 k][256 cols], 512-byte rows, 32-byte unit ^= (k & 3) | ((k >> 3) & 1) << 2
//                           (ds_read_b64_tr_b16, the transposing read: dgrad weights and both wgrad operands)
// Split-K (weight gradients) adds partial tiles with float atomics; for that variant the MFMA operands are swapped so
// that consecutive lanes hold consecutive output columns (contiguous 64-byte atomic segments).
#include <cstdlib>
#include <type_traits>

#include "gemm_common.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512, NSTAGE = 2;
constexpr int OP_BYTES = 256 * 64 * 2;     // 32 KiB per operand per stage
constexpr int STAGE_BYTES = 2 * OP_BYTES;  // 64 KiB
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
constexpr int EPI_SPLITK = 100;

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

__device__ __forceinline__ int fk(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }
__device__ __forceinline__ int off_k(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int off_mn(int krow, int chunk) { return krow * 512 + (((((chunk >> 1) ^ fk(krow)) << 1) | (chunk & 1)) << 4); }

template <int LAYOUT>
__device__ __forceinline__ bf16x8 read_frag(const char* op, int base, int s, int lane) {
	if (LAYOUT == VITK_K_MAJOR) {
		return *reinterpret_cast<const bf16x8*>(op + off_k(base + (lane & 15), 4 * s + (lane >> 4)));
	} else {
		const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
		const int krow = 32 * s + 8 * g + q;
		const int col = base + 4 * p;
		const int off = off_mn(krow, col >> 3) + ((p & 1) << 3);
		typedef s16x4 __attribute__((address_space(3))) * lds_ptr_t;
		s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(op + off));
		s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(op + off + 4 * 512));  // k rows +4: same swizzle key
		union {
			s16x4 h[2];
			bf16x8 v;
		} u;
		u.h[0] = lo;
		u.h[1] = hi;
		return u.v;
	}
}

// LDS-DMA of one operand's 256 x 64 stage: 32 pieces of 1 KiB, 4 per wave.
template <int LAYOUT>
__device__ __forceinline__ void stage_op(char* op, const bf16_t* __restrict__ base, int64_t ld, int64_t mn0, int64_t mn_total, int k0, int wave, int lane) {
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int piece = 4 * wave + j;
		const bf16_t* src;
		if (LAYOUT == VITK_K_MAJOR) {
			const int rl = lane >> 3;  // row within the piece (8 rows of 128 bytes) == row & 7
			int64_t row = mn0 + 8 * piece + rl;
			if (row >= mn_total) row = mn_total - 1;
			const int chunk = (lane & 7) ^ rl;
			src = base + row * ld + k0 + 8 * chunk;
		} else {
			const int krow = 2 * piece + (lane >> 5);
			const int pc = lane & 31;
			const int chunk = ((((pc >> 1) ^ fk(krow)) << 1) | (pc & 1));
			int64_t col = mn0 + 8 * chunk;
			if (col > mn_total - 8) col = mn_total - 8;
			src = base + (int64_t)(k0 + krow) * ld + col;
		}
		__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(op + piece * 1024), 16, 0, 0);
	}
}

struct PArgs {
	GemmDev g;
	int tiles_m, tiles_n, splits, ksteps;  // ksteps = K-tiles (of 64) per work item
	int total_work;                        // tiles * splits
	int iters;                             // ceil(total_work / gridDim.x)
	int vec4;
};

// Position of the K-step stream: which work item and which of its K-steps.
struct Cursor {
	int it;    // iteration of the persistent loop (work item = it-th of this workgroup)
	int k;     // K-step within the item
	int64_t m0, n0;
	int kbase;  // first k of the item (split-K offset)
	bool valid;
};

__device__ __forceinline__ void cursor_load(const PArgs& a, Cursor& c) {
	// workgroups with equal (blockIdx.x & 7) share an XCD (and its L2) under round-robin dispatch: give each XCD a
	// contiguous run of the tile raster per iteration
	const int G = gridDim.x;
	const int per_xcd = G >> 3;
	const int lin = (c.it * 8 + (blockIdx.x & 7)) * per_xcd + (blockIdx.x >> 3);
	c.valid = c.it < a.iters && lin < a.total_work;
	const int ntiles = a.tiles_m * a.tiles_n;
	const int item = c.valid ? lin : 0;
	const int split = item / ntiles, tile = item - split * ntiles;
	constexpr int GM = 4;  // raster: groups of 4 tile rows, columns inside
	const int per_group = GM * a.tiles_n;
	const int group = tile / per_group;
	const int first_m = group * GM;
	const int gsz = min(a.tiles_m - first_m, GM);
	const int tm = first_m + (tile % per_group) % gsz;
	const int tn = (tile % per_group) / gsz;
	c.m0 = (int64_t)tm * BM;
	c.n0 = (int64_t)tn * BN;
	c.kbase = split * a.ksteps * BK;
	c.k = 0;
}

__device__ __forceinline__ void cursor_next(const PArgs& a, Cursor& c) {
	if (++c.k == a.ksteps) {
		++c.it;
		cursor_load(a, c);
	}
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

template <int EPI, int ACT, typename TOUT>
__device__ __forceinline__ void emit4(const PArgs& a, int64_t m, int64_t nb, f32x4 v) {
	const GemmDev& g = a.g;
	if (m >= g.M || nb >= g.N) return;
	if (a.vec4 && nb + 3 < g.N) {
		const float f[4] = {v[0], v[1], v[2], v[3]};
		epi_store4<bf16_t, TOUT, EPI, ACT>(g, m, nb, f);
	} else {
		if (nb + 0 < g.N) epi_store1<bf16_t, TOUT, EPI, ACT>(g, m, nb + 0, v[0]);
		if (nb + 1 < g.N) epi_store1<bf16_t, TOUT, EPI, ACT>(g, m, nb + 1, v[1]);
		if (nb + 2 < g.N) epi_store1<bf16_t, TOUT, EPI, ACT>(g, m, nb + 2, v[2]);
		if (nb + 3 < g.N) epi_store1<bf16_t, TOUT, EPI, ACT>(g, m, nb + 3, v[3]);
	}
}

template <int ALAYOUT, int BLAYOUT, typename TOUT, int EPI, int ACT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_p_kernel(PArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const GemmDev& g = a.g;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
	const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
	const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);

	Cursor ld_c, mm_c;  // DMA cursor runs 3 K-steps ahead of the MFMA cursor
	ld_c.it = 0;
	cursor_load(a, ld_c);
	mm_c = ld_c;
	if (!mm_c.valid) return;

	f32x4 acc[8][4];
#pragma unroll
	for (int i = 0; i < 8; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

	int ld_step = 0;  // global index of the next K-tile to issue (stage = index & 1)
	{
		stage_op<ALAYOUT>(smem, A, g.lda, ld_c.m0, g.M, ld_c.kbase, wave, lane);
		stage_op<BLAYOUT>(smem + OP_BYTES, B, g.ldb, ld_c.n0, g.N, ld_c.kbase, wave, lane);
		cursor_next(a, ld_c);
		ld_step = 1;
	}
	int mm_step = 0;
#pragma unroll 1
	while (mm_c.valid) {
		// stage mm_step has landed (the fence of __syncthreads drains vmcnt(0) while an LDS-DMA is in flight) and every
		// wave is done reading the other stage
		__syncthreads();
		if (ld_c.valid) {
			char* st = smem + (ld_step & 1) * STAGE_BYTES;
			stage_op<ALAYOUT>(st, A, g.lda, ld_c.m0, g.M, ld_c.kbase + ld_c.k * BK, wave, lane);
			stage_op<BLAYOUT>(st + OP_BYTES, B, g.ldb, ld_c.n0, g.N, ld_c.kbase + ld_c.k * BK, wave, lane);
			cursor_next(a, ld_c);
			++ld_step;
		}
		{
			const char* sa = smem + (mm_step & 1) * STAGE_BYTES;
			const char* sb = sa + OP_BYTES;
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
					for (int j = 0; j < 4; ++j) {
						if (EPI == EPI_SPLITK)
							acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[i], wf[j], acc[i][j], 0, 0, 0);
						else
							acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
					}
			}
		}
		++mm_step;
		// (4) last K-step of the tile: epilogue from registers
		if (mm_c.k + 1 == a.ksteps) {
			const int64_t m0 = mm_c.m0 + wm, n0 = mm_c.n0 + wn;
			if (EPI == EPI_SPLITK) {
				// acc[i][j][r]: m = 16 i + 4 (lane >> 4) + r, n = 16 j + (lane & 15): 16 consecutive columns per lane group
				float* cbase = reinterpret_cast<float*>(g.C);
				static_for<0, 8>([&](auto ic) {
					static_for<0, 4>([&](auto jc) {
						constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
						const int64_t n = n0 + 16 * j + (lane & 15);
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							const int64_t m = m0 + 16 * i + 4 * (lane >> 4) + r;
							if (m < g.M && n < g.N) atomicAdd(cbase + m * g.ldc + n, acc[i][j][r]);
						}
						acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
					});
				});
			} else {
				// acc[i][j][r]: m = 16 i + (lane & 15), n = 16 j + 4 (lane >> 4) + r
				static_for<0, 8>([&](auto ic) {
					static_for<0, 4>([&](auto jc) {
						constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
						emit4<EPI, ACT, TOUT>(a, m0 + 16 * i + (lane & 15), n0 + 16 * j + 4 * (lane >> 4), acc[i][j]);
						acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
					});
				});
			}
		}
		cursor_next(a, mm_c);
	}
}

template <int AL, int BL, typename TOUT, int EPI, int ACT>
int launch1(const PArgs& a, dim3 grid, hipStream_t st) {
	static bool attr = false;
	if (!attr) {
		hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_p_kernel<AL, BL, TOUT, EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr = true;
	}
	hipLaunchKernelGGL((gemm_bf16_p_kernel<AL, BL, TOUT, EPI, ACT>), grid, dim3(NTHREADS), LDS_BYTES, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

// Variants instantiated (what the ViT / CvT training step uses); anything else runs on the non-persistent kernels.
//   (K, K)   forward:  plain (bf16 / f32 out), bias + activation (+ pre-activation), bias + LayerScale + residual
//   (K, MN)  dgrad:    plain, activation derivative
//   (MN, MN) wgrad:    split-K atomics, plain f32
int launch_variant(const PArgs& a, const vitk_gemm_desc& d, dim3 grid, hipStream_t st) {
	const bool f32out = d.out_dtype == VITK_F32;
	constexpr int K = VITK_K_MAJOR, MN = VITK_MN_MAJOR;
	if (d.a_layout == K && d.b_layout == K) {
		if (a.splits > 1) return VITK_ERR_UNSUPPORTED;
		switch (d.epilogue) {
			case VITK_EPI_NONE: return f32out ? launch1<K, K, float, VITK_EPI_NONE, -1>(a, grid, st) : launch1<K, K, bf16_t, VITK_EPI_NONE, -1>(a, grid, st);
			case VITK_EPI_ACT:
				if (f32out) return VITK_ERR_UNSUPPORTED;
				return d.act == VITK_ACT_GELU ? launch1<K, K, bf16_t, VITK_EPI_ACT, VITK_ACT_GELU>(a, grid, st) : launch1<K, K, bf16_t, VITK_EPI_ACT, -1>(a, grid, st);
			case VITK_EPI_RESIDUAL: return f32out ? VITK_ERR_UNSUPPORTED : launch1<K, K, bf16_t, VITK_EPI_RESIDUAL, -1>(a, grid, st);
			default: return VITK_ERR_UNSUPPORTED;
		}
	}
	if (d.a_layout == K && d.b_layout == MN) {
		if (a.splits > 1 || f32out) return VITK_ERR_UNSUPPORTED;
		switch (d.epilogue) {
			case VITK_EPI_NONE: return launch1<K, MN, bf16_t, VITK_EPI_NONE, -1>(a, grid, st);
			case VITK_EPI_ACTGRAD:
				return d.act == VITK_ACT_GELU ? launch1<K, MN, bf16_t, VITK_EPI_ACTGRAD, VITK_ACT_GELU>(a, grid, st) : launch1<K, MN, bf16_t, VITK_EPI_ACTGRAD, -1>(a, grid, st);
			default: return VITK_ERR_UNSUPPORTED;
		}
	}
	if (d.a_layout == MN && d.b_layout == MN) {
		if (!f32out || d.epilogue != VITK_EPI_NONE) return VITK_ERR_UNSUPPORTED;
		if (a.splits > 1) return launch1<MN, MN, float, EPI_SPLITK, -1>(a, grid, st);
		return launch1<MN, MN, float, VITK_EPI_NONE, -1>(a, grid, st);
	}
	return VITK_ERR_UNSUPPORTED;
}

}  // namespace

bool vitk_gemm_bf16_p_ok(const vitk_gemm_desc& d) {
	if (d.patch_grid > 0 || d.K % BK != 0) return false;
	if (d.M < 256 || d.N < 256) return false;
	if (d.out_dtype == VITK_F32 && d.epilogue != VITK_EPI_NONE) return false;
	return true;
}

int vitk_gemm_bf16_p(const vitk_gemm_desc& d, hipStream_t st) {
	PArgs a{};
	a.g = make_gemm_dev(d);
	a.tiles_m = (int)vitk_cdiv(d.M, BM);
	a.tiles_n = (int)vitk_cdiv(d.N, BN);
	const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
	const int64_t ksteps = d.K / BK;
	static int num_cu = 0;
	if (num_cu == 0) {
		int dev = 0;
		hipDeviceProp_t prop;
		if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return VITK_ERR_HIP;
		num_cu = prop.multiProcessorCount;
	}
	const int grid_x = std::max(8, num_cu / 8 * 8);  // one workgroup per CU, a multiple of the 8 XCDs
	// split-K (weight gradients: few output tiles, very long K): as few K slices as fill the chip once, every
	// slice adds a full tile of float atomics.  The slices must be equal (ksteps % splits == 0).
	a.splits = 1;
	if (d.out_dtype == VITK_F32 && d.epilogue == VITK_EPI_NONE && d.bias == nullptr && d.tokens_per_image == 0 && tiles * 2 <= grid_x && ksteps >= 16) {
		int s = (int)std::min<int64_t>(grid_x / tiles, ksteps / 8);
		while (s > 1 && ksteps % s != 0) --s;
		a.splits = std::max(1, s);
	}
	a.ksteps = (int)(ksteps / a.splits);
	a.total_work = (int)(tiles * a.splits);
	a.iters = (int)vitk_cdiv(a.total_work, grid_x);
	auto al16 = [](const void* p) { return p == nullptr || vitk_aligned(p, 16); };
	a.vec4 = (d.N % 4 == 0) && (d.ldc % 4 == 0) && (d.ld_aux % 4 == 0) && (d.ld_out2 % 4 == 0) && al16(d.C) && al16(d.bias) && al16(d.gamma) && al16(d.pos) &&
		al16(d.aux) && al16(d.out2);
	if (a.splits > 1 && !(d.a_layout == VITK_MN_MAJOR && d.b_layout == VITK_MN_MAJOR)) {
		a.splits = 1;  // split-K exists for the weight-gradient layout only
		a.ksteps = (int)ksteps;
		a.total_work = (int)tiles;
		a.iters = (int)vitk_cdiv(a.total_work, grid_x);
	}
	if (a.splits > 1 && !d.accumulate) {
		hipError_t e = hipMemset2DAsync(d.C, (size_t)d.ldc * sizeof(float), 0, (size_t)d.N * sizeof(float), (size_t)d.M, st);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	}
	return launch_variant(a, d, dim3((unsigned)grid_x), st);
}
