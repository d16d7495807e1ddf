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

// ---- dropout masks --------------------------------------------------------------------------------
// nn.Dropout (layers/vit.py:17-23, layers/transformers.py:20-21) as a counter-based mask: element i of a row-major
// [M, N] tensor is kept iff 16 bits of a hash of (seed, i / 2) are >= p * 65536; elements 2j and 2j+1 share one hash.
// The streaming dropout kernels and the GEMM epilogues evaluate the SAME function, so a mask applied by one in the
// forward pass can be re-applied by the other in the backward pass from (p, seed) alone.  Kept values are scaled by
// 65536 / (65536 - thr16), the exact inverse of the keep probability of the quantised threshold.
struct DropKey {
	uint32_t k0, k1;  // the seed, mixed on the host so that consecutive seeds give unrelated masks
	uint32_t thr16;   // 0 = dropout off
	// Device word mixed into the seed when the mask is drawn, or null (vitk_dropout_epoch_set): a launch captured into a HIP
	// graph keeps the seed it was captured with, so a step that is replayed advances this word instead (once per step, before
	// its first mask) and every replay draws new masks, the forward and the backward of one step the same ones.
	const uint32_t* epoch;
};
// the word registered for the calling thread's current device (null: none)
const uint32_t* vitk_dropout_epoch_current(void);
__host__ __device__ __forceinline__ uint64_t vitk_splitmix64(uint64_t z) {
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static inline DropKey make_drop_key(float p, uint64_t seed) {
	DropKey k{0u, 0u, 0u, nullptr};
	if (p > 0.f) {
		k.epoch = vitk_dropout_epoch_current();
		const uint64_t z = vitk_splitmix64(seed);
		k.k0 = (uint32_t)z;
		k.k1 = (uint32_t)(z >> 32);
		const float t = p * 65536.0f + 0.5f;
		k.thr16 = t >= 65535.f ? 65535u : (t < 1.f ? 1u : (uint32_t)t);
	}
	return k;
}
static inline float drop_key_scale(const DropKey& k) { return 65536.0f / (float)(65536u - k.thr16); }
// 32 mask bits for elements 2 * pair and 2 * pair + 1 (low / high half)
// split in two so that callers walking consecutive pairs whose upper index word cannot change hoist the first part
__device__ __forceinline__ uint32_t drop_epoch_term(const uint32_t* epoch) { return epoch != nullptr ? *epoch * 0x85EBCA6Bu : 0u; }
// Folds the current value of the epoch word into the key.  Every kernel calls this ONCE, at its entry, on its own copy of
// the key (its by-value argument block): the masks themselves are then drawn from registers only.
__device__ __forceinline__ void drop_key_enter(DropKey& k) {
	if (k.thr16 != 0u && k.epoch != nullptr) k.k1 += drop_epoch_term(k.epoch);
}
__device__ __forceinline__ uint32_t drop_hi_term(const DropKey& k, uint64_t pair) { return k.k1 + (uint32_t)(pair >> 32) * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t drop_pair_bits_lo(const DropKey& k, uint32_t hi_term, uint32_t pair_lo) {
	uint32_t x = pair_lo ^ k.k0;
	x ^= x >> 16;
	x *= 0x7feb352du;
	x += hi_term;
	x ^= x >> 15;
	x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}
__device__ __forceinline__ uint32_t drop_pair_bits(const DropKey& k, uint64_t pair) {
	return drop_pair_bits_lo(k, drop_hi_term(k, pair), (uint32_t)pair);
}
__device__ __forceinline__ bool drop_keep(const DropKey& k, uint64_t i) {
	const uint32_t bits = drop_pair_bits(k, i >> 1);
	return ((i & 1) ? (bits >> 16) : (bits & 0xffffu)) >= k.thr16;
}
// v[0..N) *= mask of elements i0 .. i0 + N - 1 (i0 even, N even); kept values are NOT scaled here
template <int N>
__device__ __forceinline__ void drop_mask_n(const DropKey& k, uint64_t i0, float (&v)[N]) {
#pragma unroll
	for (int j = 0; j < N / 2; ++j) {
		const uint32_t bits = drop_pair_bits(k, (i0 >> 1) + j);
		v[2 * j] = (bits & 0xffffu) >= k.thr16 ? v[2 * j] : 0.f;
		v[2 * j + 1] = (bits >> 16) >= k.thr16 ? v[2 * j + 1] : 0.f;
	}
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

// Phi(u) for results that are rounded to bf16 anyway: 0.5 + uc * Q(uc^2 / c^2) with uc = u clamped to +-c, c = 4.25
// (|error| <= 1.2e-5 absolute, 1 - Phi(c) = 1.1e-5).  No transcendental instruction (those issue at quarter rate):
// a clamp and 11 multiply-adds, which the compiler packs two elements per instruction.
__device__ __forceinline__ float gelu_cdf_poly(float u) {
	const float uc = __builtin_amdgcn_fmed3f(u, -4.25f, 4.25f);
	const float t = uc * uc * (1.0f / (4.25f * 4.25f));
	float q = fmaf(6.304869652e-01f, t, -3.341864109e+00f);
	q = fmaf(q, t, 7.832413673e+00f);
	q = fmaf(q, t, -1.081737137e+01f);
	q = fmaf(q, t, 9.943633080e+00f);
	q = fmaf(q, t, -6.532705307e+00f);
	q = fmaf(q, t, 3.202498674e+00f);
	q = fmaf(q, t, -1.198347330e+00f);
	q = fmaf(q, t, 3.989023268e-01f);
	return fmaf(uc, q, 0.5f);
}
__device__ __forceinline__ float gelu_bf16(float u) { return u * gelu_cdf_poly(u); }
__device__ __forceinline__ float gelu_deriv_bf16(float u) { return fmaf(u * 0.39894228040143268f, __expf(-0.5f * u * u), gelu_cdf_poly(u)); }

// The same for N elements at once, one polynomial step over all elements at a time: consecutive instructions are then
// independent (written element by element, each packed multiply-add waits on the one before it).
template <int N>
__device__ __forceinline__ void gelu_cdf_poly_n(const float (&u)[N], float (&cdf)[N]) {
	constexpr float C[9] = {3.989023268e-01f, -1.198347330e+00f, 3.202498674e+00f, -6.532705307e+00f, 9.943633080e+00f,
		-1.081737137e+01f, 7.832413673e+00f, -3.341864109e+00f, 6.304869652e-01f};
	float uc[N], t[N], q[N];
#pragma unroll
	for (int e = 0; e < N; ++e) uc[e] = __builtin_amdgcn_fmed3f(u[e], -4.25f, 4.25f);
#pragma unroll
	for (int e = 0; e < N; ++e) t[e] = uc[e] * uc[e] * (1.0f / (4.25f * 4.25f));
#pragma unroll
	for (int e = 0; e < N; ++e) q[e] = fmaf(C[8], t[e], C[7]);
#pragma unroll
	for (int c = 6; c >= 0; --c) {
#pragma unroll
		for (int e = 0; e < N; ++e) q[e] = fmaf(q[e], t[e], C[c]);
	}
#pragma unroll
	for (int e = 0; e < N; ++e) cdf[e] = fmaf(uc[e], q[e], 0.5f);
}
template <int N>
__device__ __forceinline__ void gelu_bf16_n(float (&v)[N]) {
	float cdf[N];
	gelu_cdf_poly_n<N>(v, cdf);
#pragma unroll
	for (int e = 0; e < N; ++e) v[e] *= cdf[e];
}
// v[e] *= gelu'(u[e])
template <int N>
__device__ __forceinline__ void gelu_deriv_bf16_n(float (&v)[N], const float (&u)[N]) {
	float cdf[N], ex[N];
#pragma unroll
	for (int e = 0; e < N; ++e) ex[e] = __builtin_amdgcn_exp2f(u[e] * u[e] * (-0.5f * 1.44269504088896341f));
	gelu_cdf_poly_n<N>(u, cdf);
#pragma unroll
	for (int e = 0; e < N; ++e) v[e] *= fmaf(u[e] * 0.39894228040143268f, ex[e], cdf[e]);
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

// Zero fill of a small float buffer as a KERNEL.  (hipMemsetAsync captured into a HIP graph as a memset node was seen to
// stop taking effect in replays that follow a hipStreamSynchronize / hipDeviceSynchronize on ROCm 7.0: accumulators that a
// captured step clears this way -- BatchNorm statistics and gradients, depthwise-convolution weight gradients -- then keep
// their values from replay to replay.  A kernel node does not have that problem.)
static __global__ void vitk_zero_f32_kernel(float* __restrict__ p, int64_t n) {
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
static inline hipError_t vitk_zero_f32(float* p, int64_t n, hipStream_t st) {
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(vitk_zero_f32_kernel, dim3(vitk_stream_grid(n, 256)), dim3(256), 0, st, p, n);
	return hipGetLastError();
}
static __global__ void vitk_zero_f32_2d_kernel(float* __restrict__ p, int64_t rows, int64_t cols, int64_t ld) {
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += (int64_t)gridDim.x * blockDim.x) p[(i / cols) * ld + i % cols] = 0.f;
}
// rows x cols floats with row stride ld (same reason as vitk_zero_f32)
static inline hipError_t vitk_zero_f32_2d(float* p, int64_t rows, int64_t cols, int64_t ld, hipStream_t st) {
	if (rows <= 0 || cols <= 0) return hipSuccess;
	hipLaunchKernelGGL(vitk_zero_f32_2d_kernel, dim3(vitk_stream_grid(rows * cols, 256)), dim3(256), 0, st, p, rows, cols, ld);
	return hipGetLastError();
}
