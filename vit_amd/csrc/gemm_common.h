// Device-side GEMM descriptor and the fused epilogue shared by the generic (float32-MFMA) and the
// bf16-MFMA GEMM kernels.
#pragma once

#include "vitk_common.h"

struct GemmDev {
	const void* A;
	const void* B;
	void* C;
	int64_t M, N, K;
	int64_t lda, ldb, ldc;
	int a_layout, b_layout;
	int epilogue, act, accumulate;
	int patch_grid, img_c, img_hw, patch_px;  // patch_px = img_hw / patch_grid
	int tpi;                                  // tokens_per_image
	int lead;                                 // free rows in front of every image (class token, distillation token), >= 1
	const float* bias;
	const float* gamma;
	const float* pos;
	const void* aux;
	int64_t ld_aux;
	void* out2;
	int64_t ld_out2;
	DropKey drop[2];   // fused dropout masks (thr16 == 0: off)
	float drop_scale;  // product of the active masks' 1 / keep-probability
	int64_t drop_m0;   // row offset of the mask index (this problem is a row block of a larger tensor)
};

static inline GemmDev make_gemm_dev(const vitk_gemm_desc& d) {
	GemmDev g{};
	g.A = d.A; g.B = d.B; g.C = d.C;
	g.M = d.M; g.N = d.N; g.K = d.K;
	g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
	g.a_layout = d.a_layout; g.b_layout = d.b_layout;
	g.epilogue = d.epilogue; g.act = d.act; g.accumulate = d.accumulate;
	g.patch_grid = d.patch_grid; g.img_c = d.img_c; g.img_hw = d.img_hw;
	g.patch_px = d.patch_grid > 0 ? d.img_hw / d.patch_grid : 0;
	g.tpi = d.tokens_per_image;
	g.lead = d.lead_rows > 1 ? d.lead_rows : 1;
	g.bias = d.bias; g.gamma = d.gamma; g.pos = d.pos;
	g.aux = d.aux; g.ld_aux = d.ld_aux; g.out2 = d.out2; g.ld_out2 = d.ld_out2;
	g.drop_scale = 1.f;
	g.drop_m0 = d.drop_row0;
	int nd = 0;
	for (int i = 0; i < 2; ++i) {
		if (d.drop_p[i] > 0.f) {  // active masks are packed to the front: drop[0].thr16 == 0 means none
			g.drop[nd] = make_drop_key(d.drop_p[i], d.drop_seed[i]);
			g.drop_scale *= drop_key_scale(g.drop[nd]);
			++nd;
		}
	}
	return g;
}

// Element offset of A(m,k) inside the NCHW image batch (im2col-free patch gather; layers/embeddings.py:12-16).
__device__ __forceinline__ int64_t patch_offset(const GemmDev& g, int64_t m, int k) {
	const int P = g.patch_grid, p = g.patch_px, hw = g.img_hw;
	const int P2 = P * P;
	const int64_t b = m / P2;
	const int t = (int)(m - b * P2);
	const int py = t / P, px = t - py * P;
	const int pp = p * p;
	const int c = k / pp, rem = k - c * pp;
	const int i = rem / p, j = rem - i * p;
	return ((b * g.img_c + c) * hw + (py * p + i)) * (int64_t)hw + (px * p + j);
}

// once per kernel, on the kernel's own argument block, before the first mask (see drop_key_enter)
__device__ __forceinline__ void gemm_dev_enter(GemmDev& g) {
	drop_key_enter(g.drop[0]);
	drop_key_enter(g.drop[1]);
}

__device__ __forceinline__ int64_t out_row(const GemmDev& g, int64_t m) { return g.tpi > 0 ? m + (m / g.tpi + 1) * g.lead : m; }

// Fused dropout: multipliers (0 or g.drop_scale) of the N consecutive elements (m, n0 .. n0 + N - 1); the mask is
// indexed by the logical element m * g.N + n.  Call only when g.drop[0].thr16 != 0.
template <int N>
__device__ __forceinline__ void drop_keep_n(const GemmDev& g, int64_t m, int64_t n0, float (&keep)[N]) {
	const uint64_t i0 = (uint64_t)((m + g.drop_m0) * g.N + n0);
	if ((N & (N - 1)) == 0 && N >= 2 && (i0 & (N - 1)) == 0) {
		// an aligned group of N: one hash per two elements and mask, and pair + j cannot carry into the upper index word
		const uint64_t pair0 = i0 >> 1;
		const uint32_t p0 = (uint32_t)pair0;
		const uint32_t h0 = drop_hi_term(g.drop[0], pair0), h1 = drop_hi_term(g.drop[1], pair0);
#pragma unroll
		for (int j = 0; j < N / 2; ++j) {
			const uint32_t b0 = drop_pair_bits_lo(g.drop[0], h0, p0 + j);
			bool lo = (b0 & 0xffffu) >= g.drop[0].thr16, hi = (b0 >> 16) >= g.drop[0].thr16;
			if (g.drop[1].thr16) {
				const uint32_t b1 = drop_pair_bits_lo(g.drop[1], h1, p0 + j);
				lo = lo && (b1 & 0xffffu) >= g.drop[1].thr16;
				hi = hi && (b1 >> 16) >= g.drop[1].thr16;
			}
			keep[2 * j] = lo ? g.drop_scale : 0.f;
			keep[2 * j + 1] = hi ? g.drop_scale : 0.f;
		}
	} else {
#pragma unroll
		for (int j = 0; j < N; ++j)
			keep[j] = drop_keep(g.drop[0], i0 + j) && (!g.drop[1].thr16 || drop_keep(g.drop[1], i0 + j)) ? g.drop_scale : 0.f;
	}
}

// Epilogue for ONE element.  EPI >= 0 fixes the epilogue kind at compile time (dead branches vanish), -1 reads g.epilogue.
template <typename TIN, typename TOUT, int EPI = -1, int ACT = -1>
__device__ __forceinline__ void epi_store1(const GemmDev& g, int64_t m, int64_t n, float acc) {
	const int epilogue = EPI >= 0 ? EPI : g.epilogue;
	const int act = ACT >= 0 ? ACT : g.act;
	float v = acc;
	if (g.bias) v += g.bias[n];
	if (g.tpi > 0 && g.pos) v += g.pos[(g.lead + m % g.tpi) * g.N + n];
	const int64_t r = out_row(g, m);
	// fused dropout: multiplier of the masked quantity (the mask is indexed by the logical element m * N + n)
	float keep = 1.f;
	if (g.drop[0].thr16) {
		const uint64_t i = (uint64_t)((m + g.drop_m0) * g.N + n);
		keep = drop_keep(g.drop[0], i) && (!g.drop[1].thr16 || drop_keep(g.drop[1], i)) ? g.drop_scale : 0.f;
	}
	if (epilogue == VITK_EPI_ACT) {
		if (g.out2) reinterpret_cast<TOUT*>(g.out2)[r * g.ld_out2 + n] = from_f32<TOUT>(v);
		// the activation sees the pre-activation as it is stored (rounded to the output type)
		v = keep * act_apply(g.out2 ? round_to<TOUT>(v) : v, act);
	} else if (epilogue == VITK_EPI_RESIDUAL) {
		const float res = to_f32<TIN>(reinterpret_cast<const TIN*>(g.aux)[r * g.ld_aux + n]);
		v = res + (g.gamma ? g.gamma[n] : 1.f) * (keep * v);
	} else if (epilogue == VITK_EPI_ACTGRAD) {
		const float u = to_f32<TIN>(reinterpret_cast<const TIN*>(g.aux)[r * g.ld_aux + n]);
		v = (keep * v) * act_deriv(u, act);
	}
	TOUT* c = reinterpret_cast<TOUT*>(g.C) + r * g.ldc + n;
	if (g.accumulate) v += to_f32<TOUT>(*c);
	*c = from_f32<TOUT>(v);
}

// Epilogue for FOUR consecutive columns n0..n0+3 of row m (all in range, 8/16-byte aligned accesses).
template <typename TIN, typename TOUT, int EPI = -1, int ACT = -1>
__device__ __forceinline__ void epi_store4(const GemmDev& g, int64_t m, int64_t n0, const float (&acc)[4]) {
	const int epilogue = EPI >= 0 ? EPI : g.epilogue;
	const int act = ACT >= 0 ? ACT : g.act;
	float v[4] = {acc[0], acc[1], acc[2], acc[3]};
	if (g.bias) {
		float b[4];
		load_vec<float, 4>(g.bias + n0, b);
#pragma unroll
		for (int j = 0; j < 4; ++j) v[j] += b[j];
	}
	if (g.tpi > 0 && g.pos) {
		float p[4];
		load_vec<float, 4>(g.pos + (g.lead + m % g.tpi) * g.N + n0, p);
#pragma unroll
		for (int j = 0; j < 4; ++j) v[j] += p[j];
	}
	const int64_t r = out_row(g, m);
	float keep[4] = {1.f, 1.f, 1.f, 1.f};  // fused dropout (see epi_store1)
	if (g.drop[0].thr16) drop_keep_n<4>(g, m, n0, keep);
	if (epilogue == VITK_EPI_ACT) {
		if (g.out2) {
			store_vec<TOUT, 4>(reinterpret_cast<TOUT*>(g.out2) + r * g.ld_out2 + n0, v);
#pragma unroll
			for (int j = 0; j < 4; ++j) v[j] = round_to<TOUT>(v[j]);
		}
#pragma unroll
		for (int j = 0; j < 4; ++j) v[j] = keep[j] * act_apply(v[j], act);
	} else if (epilogue == VITK_EPI_RESIDUAL) {
		float res[4], gm[4] = {1.f, 1.f, 1.f, 1.f};
		load_vec<TIN, 4>(reinterpret_cast<const TIN*>(g.aux) + r * g.ld_aux + n0, res);
		if (g.gamma) load_vec<float, 4>(g.gamma + n0, gm);
#pragma unroll
		for (int j = 0; j < 4; ++j) v[j] = res[j] + gm[j] * (keep[j] * v[j]);
	} else if (epilogue == VITK_EPI_ACTGRAD) {
		float u[4];
		load_vec<TIN, 4>(reinterpret_cast<const TIN*>(g.aux) + r * g.ld_aux + n0, u);
#pragma unroll
		for (int j = 0; j < 4; ++j) v[j] = (keep[j] * v[j]) * act_deriv(u[j], act);
	}
	TOUT* c = reinterpret_cast<TOUT*>(g.C) + r * g.ldc + n0;
	if (g.accumulate) {
		float o[4];
		load_vec<TOUT, 4>(c, o);
#pragma unroll
		for (int j = 0; j < 4; ++j) v[j] += o[j];
	}
	store_vec<TOUT, 4>(c, v);
}

extern thread_local const char* vitk_last_gemm_kernel;  // set by every GEMM launcher
int vitk_gemm_generic(const vitk_gemm_desc& d, hipStream_t st);
// *colsum_done is set when the kernel that ran also accumulated d.a_colsum
int vitk_gemm_bf16(const vitk_gemm_desc& d, hipStream_t st, bool* colsum_done);
bool vitk_gemm_bf16_ok(const vitk_gemm_desc& d);
int vitk_gemm_bf16_128(const vitk_gemm_desc& d, hipStream_t st, bool* colsum_done = nullptr);  // the 128x128 kernel, whatever the shape
// several weight-gradient-shaped products in one launch of the 128x128 kernel (K cut through one shared scratch buffer)
bool vitk_gemm_bf16_group_ok(const vitk_gemm_desc& d);
int vitk_gemm_group_max();
int vitk_gemm_bf16_group128(const vitk_gemm_desc* const* ds, int n, hipStream_t st);
bool vitk_gemm_bf16_256_ok(const vitk_gemm_desc& d);  // large-tile LDS-DMA kernel (needs K % 64 == 0, M, N >= 256)
int vitk_gemm_bf16_256(const vitk_gemm_desc& d, hipStream_t st, bool* colsum_done);
bool vitk_gemm_bf16_p8_ok(const vitk_gemm_desc& d);  // persistent 256x256 kernel with the deep operand pipeline (whole tiles, K-major x K-major, bf16 out)
int vitk_gemm_bf16_p8(const vitk_gemm_desc& d, hipStream_t st);
// Weight-gradient form of the same pipeline: both operands MN-major, float32 K-split partial tiles (accumulator order) into
// `partials`, optional column sums of A (the bias gradient) into `cs_partials`; whole 256x256 tiles, K slices of >= 2 steps.
struct WgradArgs {
	GemmDev g;
	int tiles_m, tiles_n, splits, k_per_split;
	float* cs_partials;  // [split][tile column][row half][M] or null
	float* partials;     // [split][tile][wave][i][j][lane][4]
};
bool vitk_gemm_bf16_p8w_ok(const WgradArgs& a);
int vitk_gemm_bf16_p8w(const WgradArgs& a, hipStream_t st);
bool vitk_gemm_bf16_t128_ok(const vitk_gemm_desc& d);  // 128x256-tile kernel, two workgroups per CU (K-major x K-major, bf16 out, K % 64 == 0)
int vitk_gemm_bf16_t128(const vitk_gemm_desc& d, hipStream_t st);
