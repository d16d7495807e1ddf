// Shared device/host helpers for the vitk kernels (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "vitk.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define VITK_WAVE 64

#define VITK_CHECK_LAUNCH()                                                  \
	do {                                                                     \
		hipError_t e__ = hipGetLastError();                                  \
		if (e__ != hipSuccess) return VITK_ERR_HIP - (int)e__;               \
	} while (0)

#define VITK_REQUIRE(cond, code) \
	do {                         \
		if (!(cond)) return (code); \
	} while (0)

static inline bool vitk_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
static inline int64_t vitk_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- scalar conversions ---------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// Round a float to the storage type and back (what a later kernel will read).
template <typename T> __device__ __forceinline__ float round_to(float v) { return to_f32<T>(from_f32<T>(v)); }

// ---- 16-byte vector access: VEC elements of T --------------------------------------------------
template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<bf16_t> { static constexpr int N = 8; };

template <typename T, int N> struct alignas((sizeof(T) * N) > 16 ? 16 : (sizeof(T) * N)) Pack { T v[N]; };

template <typename T, int N> __device__ __forceinline__ void load_vec(const T* p, float (&out)[N]) {
	Pack<T, N> pk = *reinterpret_cast<const Pack<T, N>*>(p);
#pragma unroll
	for (int i = 0; i < N; ++i) out[i] = to_f32<T>(pk.v[i]);
}
template <typename T, int N> __device__ __forceinline__ void store_vec(T* p, const float (&in)[N]) {
	Pack<T, N> pk;
#pragma unroll
	for (int i = 0; i < N; ++i) pk.v[i] = from_f32<T>(in[i]);
	*reinterpret_cast<Pack<T, N>*>(p) = pk;
}

// ---- wave reductions (64 lanes) ----------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
	return v;
}

// ---- activations (layers/transformers.py:26-33; F.gelu there is the exact erf form) -----------
// Phi(u) = 0.5 * (1 + erf(u / sqrt 2)) through erfc(x) = poly(t) * exp(-x^2), t = 1 / (1 + p x), x = |u| / sqrt 2
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute, evaluated on the erfc side so that the negative tail
// keeps its relative accuracy).  One v_rcp, one v_exp and 7 FMAs instead of the ~40-instruction erff().
// `e` returns exp(-u^2 / 2), which the derivative reuses.
__device__ __forceinline__ float gelu_cdf(float u, float& e) {
	const float x = fabsf(u) * 0.70710678118654752f;
	const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));  // one v_rcp_f32 (1 ulp), not the IEEE division sequence
	e = __expf(-x * x);
	float poly = fmaf(1.061405429f, t, -1.453152027f);
	poly = fmaf(poly, t, 1.421413741f);
	poly = fmaf(poly, t, -0.284496736f);
	poly = fmaf(poly, t, 0.254829592f);
	const float half_erfc = 0.5f * poly * t * e;
	return u >= 0.f ? 1.0f - half_erfc : half_erfc;
}

__device__ __forceinline__ float act_apply(float u, int act) {
	switch (act) {
		case VITK_ACT_GELU: {
			float e;
			return u * gelu_cdf(u, e);
		}
		case VITK_ACT_SILU: return u * __builtin_amdgcn_rcpf(1.0f + __expf(-u));
		case VITK_ACT_RELU: return fmaxf(u, 0.0f);
		case VITK_ACT_RELU6: return fminf(fmaxf(u, 0.0f), 6.0f);
		default: return u;
	}
}
__device__ __forceinline__ float act_deriv(float u, int act) {
	switch (act) {
		case VITK_ACT_GELU: {
			float e;
			const float cdf = gelu_cdf(u, e);
			return fmaf(u * 0.39894228040143268f, e, cdf);
		}
		case VITK_ACT_SILU: {
			float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
			return s * (1.0f + u * (1.0f - s));
		}
		case VITK_ACT_RELU: return u > 0.0f ? 1.0f : 0.0f;
		case VITK_ACT_RELU6: return (u > 0.0f && u < 6.0f) ? 1.0f : 0.0f;
		default: return 1.0f;
	}
}

// Grid size for a grid-stride streaming kernel: enough blocks to fill 256 CUs x 8, no more.
static inline int vitk_stream_grid(int64_t work_items, int block) {
	int64_t g = vitk_cdiv(work_items, block);
	if (g < 1) g = 1;
	if (g > 2048) g = 2048;
	return (int)g;
}
