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

// LDS-DMA of one 256 x 64 operand tile: 32 pieces of 1 KiB, 4 per wave.
template <int LAYOUT>
__device__ __forceinline__ void stage_tile(char* tile, const bf16_t* __restrict__ base, int64_t ld, int64_t mn0, int64_t mn_total, int k0, int wave, int lane) {
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int piece = 4 * wave + j;
		const bf16_t* src;
		if (LAYOUT == VITK_K_MAJOR) {
			const int r = lane >> 3;  // == row & 7
			int64_t row = mn0 + 8 * piece + r;
			if (row >= mn_total) row = mn_total - 1;
			const int chunk = (lane & 7) ^ r;
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

// Compile-time loop: the body sees its index as a constant, so accumulator arrays indexed by it can never become
// runtime-indexed (which would move them to scratch memory) even when the optimizer declines to unroll a large body.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		static_for<I + 1, N>(f);
	}
}

struct BigArgs {
	GemmDev g;
	int tiles_m, tiles_n, splits, k_per_split;
	int vec4, vec8;  // epilogue operands / outputs may be accessed 4-wide (8 / 16 bytes) resp. 8-wide (16 bytes bf16)
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

template <int ALAYOUT, int BLAYOUT, typename TOUT, int EPI, int ACT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_256_kernel(BigArgs a) {
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

	stage_tile<ALAYOUT>(smem, A, g.lda, m0, g.M, k_begin, wave, lane);
	stage_tile<BLAYOUT>(smem + TILE_BYTES, B, g.ldb, n0, g.N, k_begin, wave, lane);
	__syncthreads();  // with LDS-DMA in flight hipcc drains vmcnt(0) ahead of the barrier

	for (int t = 0; t < nt; ++t) {
		const char* sa = smem + (t & 1) * STAGE_BYTES;
		const char* sb = sa + TILE_BYTES;
		if (t + 1 < nt) {
			char* da = smem + ((t + 1) & 1) * STAGE_BYTES;
			const int k0 = k_begin + (t + 1) * BK;
			stage_tile<ALAYOUT>(da, A, g.lda, m0, g.M, k0, wave, lane);
			stage_tile<BLAYOUT>(da + TILE_BYTES, B, g.ldb, n0, g.N, k0, wave, lane);
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
		__syncthreads();  // next stage landed (vmcnt(0) drained by the fence) and everyone is done with this one
	}

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
					for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], ACT >= 0 ? ACT : g.act);
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
						for (int r = 0; r < 4; ++r) v[r] *= act_deriv(x4[r], ACT >= 0 ? ACT : g.act);
					}
				}
				bf16x4 pk;
#pragma unroll
				for (int r = 0; r < 4; ++r) pk[r] = (bf16_t)v[r];
				*reinterpret_cast<bf16x4*>(img + ml * CB_LD + nl) = pk;
			});
		});
		__syncthreads();
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
	} else {
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

// Variants instantiated (what the ViT / CvT training step uses); any other combination returns VITK_ERR_UNSUPPORTED
// and runs on the 128x128 kernel with its run-time epilogue.
//   (K, K)   forward:  plain (bf16 / f32), bias + activation (+ pre-activation), bias + LayerScale + residual
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
			case VITK_EPI_NONE: return launch1<K, K, bf16_t, VITK_EPI_NONE, -1>(a, grid, st);
			case VITK_EPI_ACT:
				return d.act == VITK_ACT_GELU ? launch1<K, K, bf16_t, VITK_EPI_ACT, VITK_ACT_GELU>(a, grid, st) : launch1<K, K, bf16_t, VITK_EPI_ACT, -1>(a, grid, st);
			case VITK_EPI_RESIDUAL: return launch1<K, K, bf16_t, VITK_EPI_RESIDUAL, -1>(a, grid, st);
			default: return VITK_ERR_UNSUPPORTED;
		}
	}
	if (d.a_layout == K && d.b_layout == MN) {
		switch (d.epilogue) {
			case VITK_EPI_NONE: return launch1<K, MN, bf16_t, VITK_EPI_NONE, -1>(a, grid, st);
			case VITK_EPI_ACTGRAD:
				return d.act == VITK_ACT_GELU ? launch1<K, MN, bf16_t, VITK_EPI_ACTGRAD, VITK_ACT_GELU>(a, grid, st)
											  : launch1<K, MN, bf16_t, VITK_EPI_ACTGRAD, -1>(a, grid, st);
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

int vitk_gemm_bf16_256(const vitk_gemm_desc& d, hipStream_t st) {
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
	a.vec8 = a.vec4 && (d.N % 8 == 0) && (d.ldc % 8 == 0) && (d.ld_out2 % 8 == 0);
	if (a.splits > 1 && !d.accumulate) {
		hipError_t e = hipMemset2DAsync(d.C, (size_t)d.ldc * sizeof(float), 0, (size_t)d.N * sizeof(float), (size_t)d.M, st);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	}
	return launch_variant(a, d, dim3((unsigned)tiles, (unsigned)a.splits), st);
}
