// Generic GEMM: any strides, any alignment, float32 or bfloat16 operands, float32 accumulation on the
// exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, bit-for-bit).
// This is the float32 parity path and the path for bf16 shapes the bf16-MFMA kernel cannot take
// (unaligned rows, tiny K / N).  64x64 output tile per 256-thread workgroup, each of the 4 waves
// owns a 32x32 quadrant; operands are staged through LDS as float32 [k][m] / [k][n] so that both
// fragment reads are 32 consecutive words (conflict-free ds_read_b32).
#include "gemm_common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 32, LDS_LD = 65;

template <typename TIN>
__device__ __forceinline__ float load_a(const GemmDev& g, int64_t m, int k) {
	const TIN* A = reinterpret_cast<const TIN*>(g.A);
	if (g.patch_grid > 0) return to_f32<TIN>(A[patch_offset(g, m, k)]);
	return to_f32<TIN>(g.a_layout == VITK_K_MAJOR ? A[m * g.lda + k] : A[(int64_t)k * g.lda + m]);
}
template <typename TIN>
__device__ __forceinline__ float load_b(const GemmDev& g, int64_t n, int k) {
	const TIN* B = reinterpret_cast<const TIN*>(g.B);
	return to_f32<TIN>(g.b_layout == VITK_K_MAJOR ? B[n * g.ldb + k] : B[(int64_t)k * g.ldb + n]);
}

// k_per_split (a multiple of BK) > 0: blockIdx.z owns the K range [z * k_per_split, (z + 1) * k_per_split) and adds its
// partial sums into C with float atomics (float32 output without epilogue only; C holds the value to add to).
template <typename TIN, typename TOUT>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmDev g, int k_per_split) {
	gemm_dev_enter(g);
	__shared__ float As[BK * LDS_LD];
	__shared__ float Bs[BK * LDS_LD];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
	const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
	f32x16 acc = {};
	const int k_begin = k_per_split > 0 ? (int)blockIdx.z * k_per_split : 0;
	const int K = k_per_split > 0 ? min((int)g.K, k_begin + k_per_split) : (int)g.K;
	// thread -> element of the 64 x 32 staging tile: consecutive threads follow the contiguous index
	const bool a_kmajor = (g.a_layout == VITK_K_MAJOR) || g.patch_grid > 0;
	const bool b_kmajor = (g.b_layout == VITK_K_MAJOR);
	// the operands of the next K step are in flight (in registers) while the matrix unit works on this one
	float ra[8], rb[8];
	auto fetch = [&](int k0) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			int mm, kk;
			if (a_kmajor) { kk = tid & 31; mm = (tid >> 5) + 8 * i; } else { mm = tid & 63; kk = (tid >> 6) + 4 * i; }
			const int64_t m = m0 + mm;
			ra[i] = (m < g.M && k0 + kk < K) ? load_a<TIN>(g, m, k0 + kk) : 0.f;
			int nn, kb;
			if (b_kmajor) { kb = tid & 31; nn = (tid >> 5) + 8 * i; } else { nn = tid & 63; kb = (tid >> 6) + 4 * i; }
			const int64_t n = n0 + nn;
			rb[i] = (n < g.N && k0 + kb < K) ? load_b<TIN>(g, n, k0 + kb) : 0.f;
		}
	};
	fetch(k_begin);
	for (int k0 = k_begin; k0 < K; k0 += BK) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			int mm, kk;
			if (a_kmajor) { kk = tid & 31; mm = (tid >> 5) + 8 * i; } else { mm = tid & 63; kk = (tid >> 6) + 4 * i; }
			As[kk * LDS_LD + mm] = ra[i];
			int nn, kb;
			if (b_kmajor) { kb = tid & 31; nn = (tid >> 5) + 8 * i; } else { nn = tid & 63; kb = (tid >> 6) + 4 * i; }
			Bs[kb * LDS_LD + nn] = rb[i];
		}
		__syncthreads();
		if (k0 + BK < K) fetch(k0 + BK);
#pragma unroll
		for (int ks = 0; ks < BK; ks += 2) {
			const float a = As[(ks + (lane >> 5)) * LDS_LD + wm + (lane & 31)];
			const float b = Bs[(ks + (lane >> 5)) * LDS_LD + wn + (lane & 31)];
			acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
		}
		__syncthreads();
	}
	// D layout: column (n) = lane & 31, row (m) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
	const int64_t n = n0 + wn + (lane & 31);
	if (n < g.N) {
#pragma unroll
		for (int r = 0; r < 16; ++r) {
			const int64_t m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
			if (m >= g.M) continue;
			if constexpr (sizeof(TOUT) == 4) {
				if (k_per_split > 0) {
					atomicAdd(reinterpret_cast<float*>(g.C) + m * g.ldc + n, acc[r]);
					continue;
				}
			}
			epi_store1<TIN, TOUT>(g, m, n, acc[r]);
		}
	}
}

}  // namespace

thread_local const char* vitk_last_gemm_kernel = "";

int vitk_gemm_generic(const vitk_gemm_desc& d, hipStream_t st) {
	vitk_last_gemm_kernel = "gemm_generic_kernel";
	GemmDev g = make_gemm_dev(d);
	const int64_t tiles = vitk_cdiv(d.N, BN) * vitk_cdiv(d.M, BM), ksteps = vitk_cdiv(d.K, BK);
	VITK_REQUIRE(vitk_cdiv(d.M, BM) <= 65535, VITK_ERR_SHAPE);
	// Few output tiles over a long K (the weight gradient of a 10-class head: one tile, K = the batch): the K loop of one
	// workgroup is a chain of load -> LDS -> matrix unit steps, so the K range is cut over workgroups that add their
	// partial sums with float atomics.  bf16 operands only: the float32 path keeps its fixed summation order.
	int splits = 1, k_per_split = 0;
	if (d.in_dtype == VITK_BF16 && d.out_dtype == VITK_F32 && d.epilogue == VITK_EPI_NONE && d.bias == nullptr && d.tokens_per_image == 0 &&
		tiles <= 64 && ksteps >= 4) {
		const int64_t steps_per = std::max<int64_t>(2, vitk_cdiv(ksteps, std::max<int64_t>(1, 256 / tiles)));
		k_per_split = (int)(steps_per * BK);
		splits = (int)vitk_cdiv(ksteps, steps_per);
		if (splits <= 1) { splits = 1; k_per_split = 0; }
	}
	if (splits > 1 && !d.accumulate) {
		hipError_t e = vitk_zero_f32_2d(reinterpret_cast<float*>(d.C), d.M, d.N, d.ldc, st);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	}
	dim3 grid((unsigned)vitk_cdiv(d.N, BN), (unsigned)vitk_cdiv(d.M, BM), (unsigned)splits);
	if (d.in_dtype == VITK_F32 && d.out_dtype == VITK_F32)
		hipLaunchKernelGGL((gemm_generic_kernel<float, float>), grid, dim3(256), 0, st, g, k_per_split);
	else if (d.in_dtype == VITK_BF16 && d.out_dtype == VITK_BF16)
		hipLaunchKernelGGL((gemm_generic_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, g, k_per_split);
	else if (d.in_dtype == VITK_BF16 && d.out_dtype == VITK_F32)
		hipLaunchKernelGGL((gemm_generic_kernel<bf16_t, float>), grid, dim3(256), 0, st, g, k_per_split);
	else
		return VITK_ERR_DTYPE;
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

static int check_desc(const vitk_gemm_desc& d) {
	VITK_REQUIRE(d.A && d.B && d.C, VITK_ERR_NULL);
	VITK_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0 && d.K < (1ll << 31) && d.N < (1ll << 31), VITK_ERR_SHAPE);
	VITK_REQUIRE(d.a_layout == VITK_K_MAJOR || d.a_layout == VITK_MN_MAJOR, VITK_ERR_UNSUPPORTED);
	VITK_REQUIRE(d.b_layout == VITK_K_MAJOR || d.b_layout == VITK_MN_MAJOR, VITK_ERR_UNSUPPORTED);
	VITK_REQUIRE(d.epilogue >= VITK_EPI_NONE && d.epilogue <= VITK_EPI_ACTGRAD, VITK_ERR_UNSUPPORTED);
	VITK_REQUIRE(d.act >= VITK_ACT_NONE && d.act <= VITK_ACT_RELU6, VITK_ERR_UNSUPPORTED);
	VITK_REQUIRE(d.ldc >= d.N, VITK_ERR_SHAPE);
	if (d.patch_grid > 0) {
		VITK_REQUIRE(d.img_c > 0 && d.img_hw > 0 && d.img_hw % d.patch_grid == 0, VITK_ERR_SHAPE);
		const int64_t p = d.img_hw / d.patch_grid;
		VITK_REQUIRE(d.K == (int64_t)d.img_c * p * p, VITK_ERR_SHAPE);
		VITK_REQUIRE(d.M % ((int64_t)d.patch_grid * d.patch_grid) == 0, VITK_ERR_SHAPE);
	} else {
		VITK_REQUIRE(d.lda >= (d.a_layout == VITK_K_MAJOR ? d.K : d.M), VITK_ERR_SHAPE);
	}
	VITK_REQUIRE(d.ldb >= (d.b_layout == VITK_K_MAJOR ? d.K : d.N), VITK_ERR_SHAPE);
	if (d.tokens_per_image > 0) VITK_REQUIRE(d.M % d.tokens_per_image == 0, VITK_ERR_SHAPE);
	if (d.epilogue == VITK_EPI_RESIDUAL || d.epilogue == VITK_EPI_ACTGRAD) VITK_REQUIRE(d.aux != nullptr && d.ld_aux >= d.N, VITK_ERR_NULL);
	if (d.out2 != nullptr) VITK_REQUIRE(d.epilogue == VITK_EPI_ACT && d.ld_out2 >= d.N, VITK_ERR_UNSUPPORTED);
	if (d.accumulate) VITK_REQUIRE(d.out_dtype == VITK_F32, VITK_ERR_UNSUPPORTED);
	VITK_REQUIRE(d.in_dtype == VITK_F32 || d.in_dtype == VITK_BF16, VITK_ERR_DTYPE);
	VITK_REQUIRE(d.out_dtype == VITK_F32 || d.out_dtype == VITK_BF16, VITK_ERR_DTYPE);
	VITK_REQUIRE(!(d.in_dtype == VITK_F32 && d.out_dtype == VITK_BF16), VITK_ERR_DTYPE);
	if (d.a_colsum != nullptr) VITK_REQUIRE(d.a_layout == VITK_MN_MAJOR && d.patch_grid == 0, VITK_ERR_UNSUPPORTED);
	VITK_REQUIRE(d.drop_row0 >= 0, VITK_ERR_SHAPE);
	for (int i = 0; i < 2; ++i) {
		VITK_REQUIRE(d.drop_p[i] >= 0.f && d.drop_p[i] < 1.f, VITK_ERR_SHAPE);
		if (d.drop_p[i] > 0.f) VITK_REQUIRE(d.epilogue != VITK_EPI_NONE && d.tokens_per_image == 0 && !d.accumulate, VITK_ERR_UNSUPPORTED);
	}
	return VITK_OK;
}

extern "C" int vitk_gemm(const vitk_gemm_desc* desc, void* stream) {
	VITK_REQUIRE(desc != nullptr, VITK_ERR_NULL);
	const vitk_gemm_desc& d = *desc;
	{
		const int rc = check_desc(d);
		if (rc != VITK_OK) return rc;
	}
	hipStream_t st = (hipStream_t)stream;
	if (d.in_dtype == VITK_BF16 && vitk_gemm_bf16_ok(d)) {
		bool colsum_done = false;
		const int rc = vitk_gemm_bf16(d, st, &colsum_done);
		if (rc != VITK_OK || d.a_colsum == nullptr || colsum_done) return rc;
	} else {
		const int rc = vitk_gemm_generic(d, st);
		if (rc != VITK_OK || d.a_colsum == nullptr) return rc;
	}
	// the kernel that ran does not fuse the column sums of A: A is [K rows][M columns] here
	return vitk_colsum(d.A, d.a_colsum, d.K, d.M, d.lda, 1, d.in_dtype, stream);
}

extern "C" int vitk_gemm_group(const vitk_gemm_desc* descs, int n, void* stream) {
	VITK_REQUIRE(n >= 0, VITK_ERR_SHAPE);
	if (n == 0) return VITK_OK;
	VITK_REQUIRE(descs != nullptr, VITK_ERR_NULL);
	for (int i = 0; i < n; ++i) {  // every descriptor is checked before anything is launched
		const int rc = check_desc(descs[i]);
		if (rc != VITK_OK) return rc;
	}
	// weight-gradient-shaped products that would each get a launch of the 128 x 128 K-split kernel and share one scratch
	// buffer go out together, up to vitk_gemm_group_max() per launch; everything else one by one
	const int cap = vitk_gemm_group_max();
	const vitk_gemm_desc* batch[16];
	int nb = 0;
	auto flush = [&]() -> int {
		int rc = nb >= 2 ? vitk_gemm_bf16_group128(batch, nb, (hipStream_t)stream) : VITK_ERR_UNSUPPORTED;
		if (rc == VITK_ERR_UNSUPPORTED) {
			rc = VITK_OK;
			for (int j = 0; j < nb && rc == VITK_OK; ++j) rc = vitk_gemm(batch[j], stream);
		}
		nb = 0;
		return rc;
	};
	for (int i = 0; i < n; ++i) {
		const vitk_gemm_desc& d = descs[i];
		if (vitk_gemm_bf16_group_ok(d) && (nb == 0 || d.workspace == batch[0]->workspace)) {
			batch[nb++] = &d;
			if (nb == cap) {
				const int rc = flush();
				if (rc != VITK_OK) return rc;
			}
			continue;
		}
		const int rc = vitk_gemm(&d, stream);
		if (rc != VITK_OK) return rc;
	}
	return flush();
}

extern "C" const char* vitk_gemm_last_kernel(void) { return vitk_last_gemm_kernel; }

extern "C" int vitk_gemm_uses_bf16_mfma(const vitk_gemm_desc* desc) {
	if (desc == nullptr) return 0;
	return (desc->in_dtype == VITK_BF16 && vitk_gemm_bf16_ok(*desc)) ? 1 : 0;
}
