// Bandwidth-bound elementwise, gather and reduction kernels of the ViT/CvT hot path.
// All of them stream 16 bytes per lane when pointers and extents allow, else fall back to
// one element per lane.
#include <cstdlib>
#include "vitk_common.h"

namespace {

// ---- generic unary / binary maps -----------------------------------------------------------------
// F: float f(float a, float b, int64 index).  b is read only if B != nullptr.
template <typename TA, typename TB, typename TO, int VEC, typename F>
__global__ __launch_bounds__(256) void map2_kernel(const TA* __restrict__ a, const TB* __restrict__ b, TO* __restrict__ o, int64_t n, F f) {
	f.enter();
	const int64_t stride = (int64_t)gridDim.x * blockDim.x * VEC;
	for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; i < n; i += stride) {
		float va[VEC], vb[VEC], vo[VEC];
		load_vec<TA, VEC>(a + i, va);
		if (b != nullptr) load_vec<TB, VEC>(b + i, vb);
#pragma unroll
		for (int j = 0; j < VEC; ++j) vo[j] = f(va[j], b != nullptr ? vb[j] : 0.f, i + j);
		store_vec<TO, VEC>(o + i, vo);
	}
}

template <typename TA, typename TB, typename TO, typename F>
int launch_map2(const void* a, const void* b, void* o, int64_t n, F f, hipStream_t st) {
	if (n <= 0) return VITK_OK;
	constexpr int V = 4;
	const bool vec = (n % V == 0) && vitk_aligned(a, sizeof(TA) * V) && (b == nullptr || vitk_aligned(b, sizeof(TB) * V)) && vitk_aligned(o, sizeof(TO) * V);
	if (vec) {
		hipLaunchKernelGGL((map2_kernel<TA, TB, TO, V, F>), dim3(vitk_stream_grid(n / V, 256)), dim3(256), 0, st, (const TA*)a, (const TB*)b, (TO*)o, n, f);
	} else {
		hipLaunchKernelGGL((map2_kernel<TA, TB, TO, 1, F>), dim3(vitk_stream_grid(n, 256)), dim3(256), 0, st, (const TA*)a, (const TB*)b, (TO*)o, n, f);
	}
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

struct Ident {
	__device__ float operator()(float a, float, int64_t) const { return a; }
	__device__ void enter() {}
};
struct ActF {
	int act;
	__device__ float operator()(float a, float, int64_t) const { return act_apply(a, act); }
	__device__ void enter() {}
};
struct ActB {
	int act;
	__device__ float operator()(float dy, float x, int64_t) const { return dy * act_deriv(x, act); }
	__device__ void enter() {}
};

// ---- dropout: the shared counter-based mask (drop_keep in vitk_common.h) ---------------------------
struct DropF {
	DropKey key;
	float inv_keep;
	__device__ float operator()(float a, float, int64_t i) const { return drop_keep(key, (uint64_t)i) ? a * inv_keep : 0.f; }
	__device__ void enter() { drop_key_enter(key); }
};
static inline DropF make_drop(float p, uint64_t seed) {
	const DropKey k = make_drop_key(p, seed);
	return DropF{k, drop_key_scale(k)};
}

// y = res + x * keep1(i) * keep2(i): one or two stacked dropouts and the residual add in one pass
struct DropAddF {
	DropF d1, d2;
	bool two;
	__device__ float operator()(float a, float res, int64_t i) const {
		float v = d1(a, 0.f, i);
		if (two) v = d2(v, 0.f, i);
		return res + v;
	}
	__device__ void enter() {
		d1.enter();
		d2.enter();
	}
};

// ---- column sum: out[n] += sum_m x[m*ld + n] ------------------------------------------------------
// blockDim = (CX, RY): CX lanes own VEC consecutive columns each, RY row groups stream disjoint rows with 4
// independent 16-byte loads in flight per lane; the row groups combine through LDS and the block issues ONE float
// atomic per column, so a column sees only gridDim.y (<= 64) atomics.
template <typename T, int VEC>
__global__ __launch_bounds__(1024) void colsum_kernel(const T* __restrict__ x, float* out, int64_t M, int64_t N, int64_t ld) {
	extern __shared__ __attribute__((aligned(16))) float red[];  // [RY][CX * VEC]
	const int cx = threadIdx.x, ry = threadIdx.y, CX = blockDim.x, RY = blockDim.y;
	const int64_t col = ((int64_t)blockIdx.x * CX + cx) * VEC;
	float acc[VEC];
#pragma unroll
	for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
	if (col < N) {
		const int64_t stride = (int64_t)gridDim.y * RY;
		int64_t m = (int64_t)blockIdx.y * RY + ry;
		for (; m + 3 * stride < M; m += 4 * stride) {
			float v[4][VEC];
#pragma unroll
			for (int u = 0; u < 4; ++u) load_vec<T, VEC>(x + (m + u * stride) * ld + col, v[u]);
#pragma unroll
			for (int u = 0; u < 4; ++u)
#pragma unroll
				for (int j = 0; j < VEC; ++j) acc[j] += v[u][j];
		}
		for (; m < M; m += stride) {
			float v[VEC];
			load_vec<T, VEC>(x + m * ld + col, v);
#pragma unroll
			for (int j = 0; j < VEC; ++j) acc[j] += v[j];
		}
	}
#pragma unroll
	for (int j = 0; j < VEC; ++j) red[(ry * CX + cx) * VEC + j] = acc[j];
	__syncthreads();
	if (ry == 0 && col < N) {
#pragma unroll
		for (int j = 0; j < VEC; ++j) {
			float s = 0.f;
			for (int r = 0; r < RY; ++r) s += red[(r * CX + cx) * VEC + j];
			atomicAdd(&out[col + j], s);
		}
	}
}

template <typename T>
int launch_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ld, hipStream_t st) {
	constexpr int V = VecOf<T>::N;
	const bool vec = (N % V == 0) && (ld % V == 0) && vitk_aligned(x, 16);
	const int vw = vec ? V : 1;
	const int64_t chunks = vitk_cdiv(N, vw);
	const int cx = (int)std::min<int64_t>(128, vitk_cdiv(chunks, 32) * 32);  // lanes along columns
	const int ry = std::max(1, 1024 / cx);
	const unsigned gx = (unsigned)vitk_cdiv(chunks, cx);
	// Few workgroups along the rows: each ends with one float atomic per column, and atomics of different workgroups on
	// one address run one after the other at the memory side (about 0.2 us each: 128 of them cost more than streaming
	// the matrix).  64 (32 for a few MB) measured best from 33280 x 192 to 50432 x 3072.
	const int64_t gy_cap = M * N * (int64_t)sizeof(T) < (16 << 20) ? 32 : 64;
	const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(vitk_cdiv(M, (int64_t)ry * 4), gy_cap));
	const size_t lds = (size_t)ry * cx * vw * sizeof(float);
	if (vec)
		hipLaunchKernelGGL((colsum_kernel<T, V>), dim3(gx, gy), dim3(cx, ry), lds, st, (const T*)x, out, M, N, ld);
	else
		hipLaunchKernelGGL((colsum_kernel<T, 1>), dim3(gx, gy), dim3(cx, ry), lds, st, (const T*)x, out, M, N, ld);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

// ---- gated FFN activation ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gated_fwd_kernel(const T* __restrict__ u, T* __restrict__ a, int64_t M, int64_t F, int64_t ld_u, int act) {
	const int64_t n = M * F;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t m = i / F, f = i - m * F;
		const float gate = to_f32<T>(u[m * ld_u + f]);
		const float up = to_f32<T>(u[m * ld_u + F + f]);
		a[i] = from_f32<T>(gate * act_apply(up, act));
	}
}
template <typename T>
__global__ __launch_bounds__(256) void gated_bwd_kernel(const T* __restrict__ da, const T* __restrict__ u, T* __restrict__ du, int64_t M, int64_t F,
	int64_t ld_u, int act) {
	const int64_t n = M * F;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t m = i / F, f = i - m * F;
		const float gate = to_f32<T>(u[m * ld_u + f]);
		const float up = to_f32<T>(u[m * ld_u + F + f]);
		const float g = to_f32<T>(da[i]);
		du[m * ld_u + f] = from_f32<T>(g * act_apply(up, act));
		du[m * ld_u + F + f] = from_f32<T>(g * gate * act_deriv(up, act));
	}
}

// ---- y = res + gamma[col] * x -----------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void scale_residual_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
	T* __restrict__ y, int64_t n, int64_t d) {
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const float g = gamma ? gamma[i % d] : 1.f;
		y[i] = from_f32<T>(to_f32<T>(res[i]) + g * to_f32<T>(x[i]));
	}
}

// dbranch = gamma * dy ; dgamma[col] += sum_rows dy * branch
template <typename T>
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ branch, const float* __restrict__ gamma,
	T* __restrict__ dbranch, float* dgamma, int64_t rows, int64_t d) {
	// block x: 256 columns; block y: row slab
	const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (col >= d) return;
	const float g = gamma[col];
	float acc = 0.f;
	for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
		const float v = to_f32<T>(dy[r * d + col]);
		acc += v * to_f32<T>(branch[r * d + col]);
		dbranch[r * d + col] = from_f32<T>(g * v);
	}
	atomicAdd(&dgamma[col], acc);
}

// ---- class-token rows and positional-embedding gradient ------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cls_rows_kernel(T* x, const float* __restrict__ cls, const float* __restrict__ pos, int64_t B, int64_t tokens, int64_t d) {
	const int64_t n = B * d;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t b = i / d, c = i - b * d;
		x[b * tokens * d + c] = from_f32<T>(cls[c] + pos[c]);
	}
}

template <typename T>
__global__ __launch_bounds__(256) void batch_sum_kernel(const T* __restrict__ dx, float* dpos, int64_t B, int64_t per_image, int accumulate, int64_t bchunk) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= per_image) return;
	const int64_t b0 = (int64_t)blockIdx.y * bchunk;
	const int64_t b1 = b0 + bchunk < B ? b0 + bchunk : B;
	float acc = 0.f;
	for (int64_t b = b0; b < b1; ++b) acc += to_f32<T>(dx[b * per_image + i]);
	if (gridDim.y == 1 && !accumulate)
		dpos[i] = acc;
	else
		atomicAdd(&dpos[i], acc);
}

// ---- patch gather / scatter (layers/embeddings.py:7-16) ------------------------------------------
template <typename T, bool SCATTER>
__global__ __launch_bounds__(256) void patchify_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t B, int C, int HW, int P) {
	const int p = HW / P;
	const int64_t K = (int64_t)C * p * p;
	const int64_t n = B * P * P * K;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t m = i / K;
		const int k = (int)(i - m * K);
		const int64_t b = m / (P * P);
		const int t = (int)(m - b * P * P);
		const int py = t / P, px = t - py * P;
		const int c = k / (p * p), rem = k - c * p * p;
		const int ii = rem / p, jj = rem - ii * p;
		const int64_t img = ((b * C + c) * HW + (py * p + ii)) * HW + (px * p + jj);
		if (SCATTER)
			dst[img] = src[i];
		else
			dst[i] = src[img];
	}
}

// The same with V consecutive elements (16 bytes) of a pixel row per thread; needs p % V == 0 (a chunk stays inside one
// pixel row of one patch) and 16-byte aligned buffers.  One index decomposition per chunk instead of one per element.
template <typename T, int V, bool SCATTER>
__global__ __launch_bounds__(256) void patchify_vec_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t B, int C, int HW, int P) {
	const int p = HW / P;
	const int64_t K = (int64_t)C * p * p;
	const int64_t nchunk = B * P * P * K / V;
	for (int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunk; ch += (int64_t)gridDim.x * blockDim.x) {
		const int64_t i = ch * V;
		const int64_t m = i / K;
		const int k = (int)(i - m * K);
		const int64_t b = m / (P * P);
		const int t = (int)(m - b * P * P);
		const int py = t / P, px = t - py * P;
		const int c = k / (p * p), rem = k - c * p * p;
		const int ii = rem / p, jj = rem - ii * p;
		const int64_t img = ((b * C + c) * HW + (py * p + ii)) * HW + (px * p + jj);
		if (SCATTER)
			*reinterpret_cast<Pack<T, V>*>(dst + img) = *reinterpret_cast<const Pack<T, V>*>(src + i);
		else
			*reinterpret_cast<Pack<T, V>*>(dst + i) = *reinterpret_cast<const Pack<T, V>*>(src + img);
	}
}

template <typename T>
__global__ __launch_bounds__(256) void drop_cls_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t B, int64_t tokens, int64_t d, int64_t lead) {
	const int64_t per = (tokens - lead) * d;
	const int64_t n = B * per;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t b = i / per, r = i - b * per;
		out[i] = x[b * tokens * d + lead * d + r];
	}
}

// Classifier + distillation head: blockIdx.y = head (token row 0 / 1), one wave per class.  The wave keeps its weight row
// in registers (lane l holds elements l, l + 64, ...; up to 16 per lane, longer rows are re-read), walks the batch and
// reduces each dot product with shuffles.
template <typename T>
__global__ __launch_bounds__(256) void dual_head_kernel(const T* __restrict__ x, const T* __restrict__ w0, const float* __restrict__ b0, const T* __restrict__ w1,
	const float* __restrict__ b1, T* __restrict__ y0, T* __restrict__ y1, int64_t B, int64_t tokens, int d, int C, int64_t samples_per_wave) {
	const int lane = threadIdx.x & 63;
	const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (c >= C) return;
	const int head = blockIdx.y;
	const T* w = (head ? w1 : w0) + (int64_t)c * d;
	const float* bias = head ? b1 : b0;
	T* y = head ? y1 : y0;
	const float bv = bias ? bias[c] : 0.f;
	float wr[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) wr[i] = lane + 64 * i < d ? to_f32<T>(w[lane + 64 * i]) : 0.f;
	// blockIdx.z: a slab of samples (with few classes the batch is what fills the chip; the weight row, d values, is cheap to re-read)
	const int64_t b_begin = (int64_t)blockIdx.z * samples_per_wave;
	const int64_t b_end = b_begin + samples_per_wave < B ? b_begin + samples_per_wave : B;
	for (int64_t b = b_begin; b < b_end; ++b) {
		const T* xr = x + (b * tokens + head) * d;
		float acc = 0.f;
#pragma unroll
		for (int i = 0; i < 16; ++i)
			if (lane + 64 * i < d) acc = fmaf(to_f32<T>(xr[lane + 64 * i]), wr[i], acc);
		for (int k = lane + 1024; k < d; k += 64) acc = fmaf(to_f32<T>(xr[k]), to_f32<T>(w[k]), acc);
		acc = wave_sum(acc);
		if (lane == 0) y[b * C + c] = from_f32<T>(acc + bv);
	}
}

template <typename T>
__global__ __launch_bounds__(256) void mean_pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t B, int64_t S, int64_t d) {
	const int64_t n = B * d;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t b = i / d, c = i - b * d;
		float acc = 0.f;
		for (int64_t s = 0; s < S; ++s) acc += to_f32<T>(x[(b * S + s) * d + c]);
		y[i] = from_f32<T>(acc / (float)S);
	}
}
template <typename T>
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int64_t B, int64_t S, int64_t d) {
	const int64_t n = B * S * d;
	const float inv = 1.f / (float)S;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t b = i / (S * d), c = i % d;
		dx[i] = from_f32<T>(to_f32<T>(dy[b * d + c]) * inv);
	}
}

// ---- cross entropy (+ label smoothing) fused with the DeiT soft-distillation term -----------------
// One wavefront per sample.
template <typename T>
__global__ __launch_bounds__(256) void ce_kd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels, const T* __restrict__ teacher,
	float* loss, T* __restrict__ dlogits, int64_t B, int C, float ls, float w_ce, float alpha, float tau) {
	const int lane = threadIdx.x & 63;
	const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (row >= B) return;
	const T* s = logits + row * C;
	const T* t = teacher ? teacher + row * C : nullptr;
	const float inv_tau = 1.f / tau;
	float m1 = -INFINITY, mt = -INFINITY;
	for (int c = lane; c < C; c += 64) {
		m1 = fmaxf(m1, to_f32<T>(s[c]));
		if (t) mt = fmaxf(mt, to_f32<T>(t[c]));
	}
	m1 = wave_max(m1);
	if (t) mt = wave_max(mt);
	float z1 = 0.f, zs = 0.f, zt = 0.f, sum_s = 0.f;
	for (int c = lane; c < C; c += 64) {
		const float v = to_f32<T>(s[c]);
		z1 += expf(v - m1);
		sum_s += v;
		if (t) {
			zs += expf((v - m1) * inv_tau);
			zt += expf((to_f32<T>(t[c]) - mt) * inv_tau);
		}
	}
	z1 = wave_sum(z1);
	sum_s = wave_sum(sum_s);
	const float lse1 = m1 + logf(z1);
	float lse_s = 0.f, lse_t = 0.f;
	if (t) {
		zs = wave_sum(zs);
		zt = wave_sum(zt);
		lse_s = m1 * inv_tau + logf(zs);
		lse_t = mt * inv_tau + logf(zt);
	}
	const int y = (int)labels[row];
	const float invB = 1.f / (float)B;
	float kl = 0.f;
	for (int c = lane; c < C; c += 64) {
		const float v = to_f32<T>(s[c]);
		const float p = expf(v - lse1);
		const float target = (c == y ? 1.f - ls : 0.f) + ls / (float)C;
		float g = w_ce * (p - target) * invB;
		if (t) {
			const float lps = v * inv_tau - lse_s;
			const float lpt = to_f32<T>(t[c]) * inv_tau - lse_t;
			const float pt = expf(lpt);
			kl += pt * (lpt - lps);
			g += alpha * tau * (expf(lps) - pt) * invB;
		}
		dlogits[row * C + c] = from_f32<T>(g);
	}
	if (t) kl = wave_sum(kl);
	if (lane == 0) {
		const float nll = lse1 - to_f32<T>(s[y]);
		const float smooth = lse1 - sum_s / (float)C;
		float l = w_ce * ((1.f - ls) * nll + ls * smooth);
		if (t) l += alpha * tau * tau * kl;
		atomicAdd(loss, l * invB);
	}
}

// ---- fused AdamW -----------------------------------------------------------------------------------
// step_state (nullable): two int64 words in device memory, [0] = step count, [1] = the step's two bias-correction factors
// as floats (computed in double, like the host path), so that a captured (graph-replayed) launch sees a new step each time.
// hp (nullable): hyper-parameters in device memory, {lr, weight_decay} of the decayed group, {lr, weight_decay} of the other
// group, beta1, beta2, eps (8 floats with one pad); they override the by-value arguments, so a captured launch follows a
// learning-rate schedule.
__global__ void adamw_advance_kernel(int64_t* step_state, float b1, float b2, const float* hp) {
	if (hp != nullptr) {
		b1 = hp[4];
		b2 = hp[5];
	}
	const int64_t st = step_state[0] + 1;
	step_state[0] = st;
	float* corr = reinterpret_cast<float*>(step_state + 1);  // [1]: the two bias-correction factors of this step
	corr[0] = (float)(1.0 / (1.0 - pow((double)b1, (double)st)));
	corr[1] = (float)(1.0 / sqrt(1.0 - pow((double)b2, (double)st)));
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* g, float* __restrict__ m, float* __restrict__ v,
	const uint8_t* __restrict__ wd_mask, bf16_t* __restrict__ p16, int64_t n, float lr, float b1, float b2, float eps, float wd, float inv_bc1,
	float inv_sqrt_bc2, float gscale, const int64_t* __restrict__ step_state, const float* __restrict__ hp, float* g_clear = nullptr) {
	if (step_state != nullptr) {
		const float* corr = reinterpret_cast<const float*>(step_state + 1);
		inv_bc1 = corr[0];
		inv_sqrt_bc2 = corr[1];
	}
	float lr1 = lr, wd1 = 0.f;  // the group wd_mask leaves out
	if (hp != nullptr) {
		lr = hp[0];
		wd = hp[1];
		lr1 = hp[2];
		wd1 = hp[3];
		b1 = hp[4];
		b2 = hp[5];
		eps = hp[6];
	}
	const float lr0 = lr, wd0 = wd;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const float gr = g[i] * gscale;
		float pv = p[i];
		const bool decay = wd_mask ? wd_mask[i] != 0 : true;
		lr = decay ? lr0 : lr1;
		pv *= 1.f - lr * (decay ? wd0 : wd1);
		const float mv = b1 * m[i] + (1.f - b1) * gr;
		const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
		const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
		pv -= lr * inv_bc1 * mv / denom;
		p[i] = pv;
		m[i] = mv;
		v[i] = vv;
		if (p16) p16[i] = (bf16_t)pv;
		if (g_clear) g_clear[i] = 0.f;  // (g_clear == g: the gradient is consumed)
	}
}

}  // namespace

#define DISPATCH_T(dtype, ...)                  \
	do {                                        \
		if ((dtype) == VITK_F32) {              \
			using T = float;                    \
			__VA_ARGS__;                        \
		} else if ((dtype) == VITK_BF16) {      \
			using T = bf16_t;                   \
			__VA_ARGS__;                        \
		} else                                  \
			return VITK_ERR_DTYPE;              \
	} while (0)

// y[c][r] = x[r][c]: 64x64 tiles through LDS (row pitch 65 elements: conflict-free both ways), 256 threads.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy) {
	__shared__ T tile[64][65];
	const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
	for (int i = ty; i < 64; i += 4) {
		const int64_t r = r0 + i, c = c0 + tx;
		if (r < rows && c < cols) tile[i][tx] = x[r * ldx + c];
	}
	__syncthreads();
#pragma unroll 4
	for (int i = ty; i < 64; i += 4) {
		const int64_t c = c0 + i, r = r0 + tx;
		if (r < rows && c < cols) y[c * ldy + r] = tile[tx][i];
	}
}

// Many transposes in one launch: block b works on tile (tiles[b].y, tiles[b].z) of matrix tiles[b].x.
template <typename T>
__global__ __launch_bounds__(256) void transpose_batched_kernel(const vitk_transpose_item* __restrict__ items, const int4* __restrict__ tiles) {
	__shared__ T tile[64][65];
	const int4 tb = tiles[blockIdx.x];
	const vitk_transpose_item it = items[tb.x];
	const T* __restrict__ x = reinterpret_cast<const T*>(it.src);
	T* __restrict__ y = reinterpret_cast<T*>(it.dst);
	const int64_t r0 = (int64_t)tb.y * 64, c0 = (int64_t)tb.z * 64;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
	for (int i = ty; i < 64; i += 4) {
		const int64_t r = r0 + i, c = c0 + tx;
		if (r < it.rows && c < it.cols) tile[i][tx] = x[r * it.cols + c];
	}
	__syncthreads();
#pragma unroll 4
	for (int i = ty; i < 64; i += 4) {
		const int64_t c = c0 + i, r = r0 + tx;
		if (r < it.rows && c < it.cols) y[c * it.rows + r] = tile[tx][i];
	}
}

extern "C" int vitk_transpose_batched(const vitk_transpose_item* items_dev, const int32_t* tiles_dev, int64_t n_tiles, int dtype, void* stream) {
	VITK_REQUIRE(items_dev && tiles_dev, VITK_ERR_NULL);
	VITK_REQUIRE(n_tiles >= 0 && n_tiles < (1ll << 31), VITK_ERR_SHAPE);
	if (n_tiles == 0) return VITK_OK;
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((transpose_batched_kernel<T>), dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, items_dev, reinterpret_cast<const int4*>(tiles_dev));
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_transpose(const void* x, void* y, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy, int dtype, void* stream) {
	VITK_REQUIRE(x && y, VITK_ERR_NULL);
	VITK_REQUIRE(rows >= 0 && cols >= 0 && ldx >= cols && ldy >= rows, VITK_ERR_SHAPE);
	if (rows == 0 || cols == 0) return VITK_OK;
	const int64_t gx = vitk_cdiv(cols, 64), gy = vitk_cdiv(rows, 64);
	VITK_REQUIRE(gx < (1ll << 31) && gy < 65536, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((transpose_kernel<T>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, rows, cols, ldx, ldy);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

struct ScaleByDev {
	const float* g;
	__device__ float operator()(float a, float, int64_t) const { return a * g[0]; }
	__device__ void enter() {}
};

extern "C" int vitk_scale_by(const void* x, const float* g, void* y, int64_t n, int dtype, void* stream) {
	VITK_REQUIRE(x && g && y, VITK_ERR_NULL);
	VITK_REQUIRE(n >= 0, VITK_ERR_SHAPE);
	hipStream_t st = (hipStream_t)stream;
	if (dtype == VITK_F32) return launch_map2<float, float, float>(x, nullptr, y, n, ScaleByDev{g}, st);
	if (dtype == VITK_BF16) return launch_map2<bf16_t, float, bf16_t>(x, nullptr, y, n, ScaleByDev{g}, st);
	return VITK_ERR_DTYPE;
}

extern "C" int vitk_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream) {
	VITK_REQUIRE(src && dst, VITK_ERR_NULL);
	VITK_REQUIRE(n >= 0, VITK_ERR_SHAPE);
	hipStream_t st = (hipStream_t)stream;
	if (src_dtype == VITK_F32 && dst_dtype == VITK_BF16) return launch_map2<float, float, bf16_t>(src, nullptr, dst, n, Ident{}, st);
	if (src_dtype == VITK_BF16 && dst_dtype == VITK_F32) return launch_map2<bf16_t, float, float>(src, nullptr, dst, n, Ident{}, st);
	if (src_dtype == VITK_F32 && dst_dtype == VITK_F32) return launch_map2<float, float, float>(src, nullptr, dst, n, Ident{}, st);
	if (src_dtype == VITK_BF16 && dst_dtype == VITK_BF16) return launch_map2<bf16_t, float, bf16_t>(src, nullptr, dst, n, Ident{}, st);
	return VITK_ERR_DTYPE;
}

extern "C" int vitk_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, void* stream) {
	VITK_REQUIRE(x && y, VITK_ERR_NULL);
	VITK_REQUIRE(act >= 0 && act <= VITK_ACT_RELU6, VITK_ERR_UNSUPPORTED);
	DISPATCH_T(dtype, return (launch_map2<T, T, T>(x, nullptr, y, n, ActF{act}, (hipStream_t)stream)));
}

extern "C" int vitk_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int act, int dtype, void* stream) {
	VITK_REQUIRE(dy && x && dx, VITK_ERR_NULL);
	VITK_REQUIRE(act >= 0 && act <= VITK_ACT_RELU6, VITK_ERR_UNSUPPORTED);
	DISPATCH_T(dtype, return (launch_map2<T, T, T>(dy, x, dx, n, ActB{act}, (hipStream_t)stream)));
}

extern "C" int vitk_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
	VITK_REQUIRE(x && y, VITK_ERR_NULL);
	VITK_REQUIRE(n >= 0 && p >= 0.f && p < 1.f, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, return (launch_map2<T, T, T>(x, nullptr, y, n, make_drop(p, seed), (hipStream_t)stream)));
}

extern "C" int vitk_dropout_add(const void* x, const void* res, void* y, int64_t n, float p1, uint64_t seed1, float p2, uint64_t seed2, int dtype,
	void* stream) {
	VITK_REQUIRE(x && y, VITK_ERR_NULL);
	VITK_REQUIRE(n >= 0 && p1 >= 0.f && p1 < 1.f && p2 >= 0.f && p2 < 1.f, VITK_ERR_SHAPE);
	const DropAddF f{make_drop(p1, seed1), make_drop(p2, seed2), p2 > 0.f};
	DISPATCH_T(dtype, return (launch_map2<T, T, T>(x, res, y, n, f, (hipStream_t)stream)));
}

extern "C" int vitk_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ld, int accumulate, int dtype, void* stream) {
	VITK_REQUIRE(x && out, VITK_ERR_NULL);
	VITK_REQUIRE(M > 0 && N > 0 && ld >= N, VITK_ERR_SHAPE);
	hipStream_t st = (hipStream_t)stream;
	if (!accumulate) {
		hipError_t e = vitk_zero_f32(out, N, st);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	}
	DISPATCH_T(dtype, return launch_colsum<T>(x, out, M, N, ld, st));
}

extern "C" int vitk_gated_act_fwd(const void* u, void* a, int64_t M, int64_t F, int64_t ld_u, int act, int dtype, void* stream) {
	VITK_REQUIRE(u && a, VITK_ERR_NULL);
	VITK_REQUIRE(M > 0 && F > 0 && ld_u >= 2 * F, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((gated_fwd_kernel<T>), dim3(vitk_stream_grid(M * F, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)u, (T*)a, M, F, ld_u, act);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_gated_act_bwd(const void* da, const void* u, void* du, int64_t M, int64_t F, int64_t ld_u, int act, int dtype, void* stream) {
	VITK_REQUIRE(da && u && du, VITK_ERR_NULL);
	VITK_REQUIRE(M > 0 && F > 0 && ld_u >= 2 * F, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((gated_bwd_kernel<T>), dim3(vitk_stream_grid(M * F, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)da, (const T*)u, (T*)du, M, F, ld_u,
			act);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_scale_residual(const void* x, const void* res, const float* gamma, void* y, int64_t rows, int64_t d, int dtype, void* stream) {
	VITK_REQUIRE(x && res && y, VITK_ERR_NULL);
	VITK_REQUIRE(rows > 0 && d > 0, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((scale_residual_kernel<T>), dim3(vitk_stream_grid(rows * d, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)res, gamma,
			(T*)y, rows * d, d);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_layerscale_bwd(const void* dy, const void* branch, const float* gamma, void* dbranch, float* dgamma, int64_t rows, int64_t d,
	int dtype, void* stream) {
	VITK_REQUIRE(dy && branch && gamma && dbranch && dgamma, VITK_ERR_NULL);
	VITK_REQUIRE(rows > 0 && d > 0, VITK_ERR_SHAPE);
	const unsigned gx = (unsigned)vitk_cdiv(d, 256);
	const unsigned gy = (unsigned)std::min<int64_t>(rows, std::max<int64_t>(1, 1024 / gx));
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((layerscale_bwd_kernel<T>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)branch, gamma, (T*)dbranch, dgamma,
			rows, d);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_cls_rows(void* x, const float* class_token, const float* pos_embed, int64_t B, int64_t tokens, int64_t d, int dtype, void* stream) {
	VITK_REQUIRE(x && class_token && pos_embed, VITK_ERR_NULL);
	VITK_REQUIRE(B > 0 && tokens > 0 && d > 0, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((cls_rows_kernel<T>), dim3(vitk_stream_grid(B * d, 256)), dim3(256), 0, (hipStream_t)stream, (T*)x, class_token, pos_embed, B, tokens, d);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_batch_sum(const void* dx, float* dpos, int64_t B, int64_t rows, int64_t d, int accumulate, int dtype, void* stream) {
	VITK_REQUIRE(dx && dpos, VITK_ERR_NULL);
	VITK_REQUIRE(B > 0 && rows > 0 && d > 0, VITK_ERR_SHAPE);
	const int64_t per = rows * d;
	const unsigned gx = (unsigned)vitk_cdiv(per, 256);
	// split the batch so that the grid has ~2048 blocks
	int64_t ysplit = std::max<int64_t>(1, std::min<int64_t>(B, 2048 / std::max<unsigned>(gx, 1u)));
	const int64_t bchunk = vitk_cdiv(B, ysplit);
	ysplit = vitk_cdiv(B, bchunk);
	hipStream_t st = (hipStream_t)stream;
	if (ysplit > 1 && !accumulate) {
		hipError_t e = vitk_zero_f32(dpos, per, st);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	}
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((batch_sum_kernel<T>), dim3(gx, (unsigned)ysplit), dim3(256), 0, st, (const T*)dx, dpos, B, per, accumulate, bchunk);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

static int patch_args_ok(int64_t B, int64_t C, int64_t HW, int64_t P) {
	if (B <= 0 || C <= 0 || HW <= 0 || P <= 0 || HW % P != 0) return 0;
	return 1;
}

extern "C" int vitk_patchify(const void* images, void* patches, int64_t B, int64_t C, int64_t HW, int64_t n_patch, int dtype, void* stream) {
	VITK_REQUIRE(images && patches, VITK_ERR_NULL);
	VITK_REQUIRE(patch_args_ok(B, C, HW, n_patch), VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		constexpr int V = 16 / (int)sizeof(T);
		if ((HW / n_patch) % V == 0 && vitk_aligned(images, 16) && vitk_aligned(patches, 16))
			hipLaunchKernelGGL((patchify_vec_kernel<T, V, false>), dim3(vitk_stream_grid(B * C * HW * HW / V, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)images,
				(T*)patches, B, (int)C, (int)HW, (int)n_patch);
		else
			hipLaunchKernelGGL((patchify_kernel<T, false>), dim3(vitk_stream_grid(B * C * HW * HW, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)images,
				(T*)patches, B, (int)C, (int)HW, (int)n_patch);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_unpatchify(const void* patches, void* images, int64_t B, int64_t C, int64_t HW, int64_t n_patch, int dtype, void* stream) {
	VITK_REQUIRE(images && patches, VITK_ERR_NULL);
	VITK_REQUIRE(patch_args_ok(B, C, HW, n_patch), VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		constexpr int V = 16 / (int)sizeof(T);
		if ((HW / n_patch) % V == 0 && vitk_aligned(images, 16) && vitk_aligned(patches, 16))
			hipLaunchKernelGGL((patchify_vec_kernel<T, V, true>), dim3(vitk_stream_grid(B * C * HW * HW / V, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)patches,
				(T*)images, B, (int)C, (int)HW, (int)n_patch);
		else
			hipLaunchKernelGGL((patchify_kernel<T, true>), dim3(vitk_stream_grid(B * C * HW * HW, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)patches, (T*)images,
				B, (int)C, (int)HW, (int)n_patch);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

namespace {
template <typename T>
__global__ void add_rows_kernel(T* __restrict__ dst, const T* __restrict__ src, int64_t rows, int64_t d, int64_t ld_dst, int64_t ld_src) {
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * d; i += (int64_t)gridDim.x * blockDim.x) {
		const int64_t r = i / d, c = i - r * d;
		dst[r * ld_dst + c] = from_f32<T>(to_f32<T>(dst[r * ld_dst + c]) + to_f32<T>(src[r * ld_src + c]));
	}
}
}  // namespace

extern "C" int vitk_add_rows(void* dst, const void* src, int64_t rows, int64_t d, int64_t ld_dst, int64_t ld_src, int dtype, void* stream) {
	VITK_REQUIRE(dst && src, VITK_ERR_NULL);
	VITK_REQUIRE(rows > 0 && d > 0 && ld_dst >= d && ld_src >= d, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((add_rows_kernel<T>), dim3(vitk_stream_grid(rows * d, 256)), dim3(256), 0, (hipStream_t)stream, (T*)dst, (const T*)src, rows, d, ld_dst, ld_src);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_drop_lead_rows(const void* x, void* out, int64_t B, int64_t tokens, int64_t d, int64_t lead, int dtype, void* stream) {
	VITK_REQUIRE(x && out, VITK_ERR_NULL);
	VITK_REQUIRE(B > 0 && lead >= 1 && tokens > lead && d > 0, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((drop_cls_kernel<T>), dim3(vitk_stream_grid(B * (tokens - lead) * d, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)out, B, tokens, d, lead);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_dual_head_fwd(const void* x, const void* w_cls, const float* b_cls, const void* w_dist, const float* b_dist, void* logits_cls, void* logits_dist,
	int64_t B, int64_t tokens, int64_t d, int64_t C, int dtype, void* stream) {
	VITK_REQUIRE(x && w_cls && w_dist && logits_cls && logits_dist, VITK_ERR_NULL);
	VITK_REQUIRE(B > 0 && tokens >= 2 && d > 0 && C > 0 && d < (1ll << 31) && C < (1ll << 31), VITK_ERR_SHAPE);
	// a wave per (head, class, slab of samples): about 8192 waves over the chip.  1000 classes: one slab (the weight matrix is
	// read once, the kernel is bound by it); 10 classes at batch 512: 410 slabs of 2 samples (one slab took 569 us there, a
	// chain of 512 dependent row reads per wave on 20 waves)
	int64_t slabs = 8192 / (2 * C);
	slabs = slabs < 1 ? 1 : (slabs > B ? B : slabs);
	if (slabs > 65535) slabs = 65535;
	const int64_t per = vitk_cdiv(B, slabs);
	slabs = vitk_cdiv(B, per);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((dual_head_kernel<T>), dim3((unsigned)vitk_cdiv(C, 4), 2, (unsigned)slabs), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)w_cls, b_cls, (const T*)w_dist, b_dist,
			(T*)logits_cls, (T*)logits_dist, B, tokens, (int)d, (int)C, per);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_deit_dual_loss(const void* logits_cls, const void* logits_dist, const int64_t* labels, const void* teacher, float* loss, void* dlogits_cls,
	void* dlogits_dist, int64_t B, int64_t C, float label_smoothing, float alpha, float tau, int dtype, void* stream) {
	VITK_REQUIRE(logits_cls && logits_dist && labels && teacher && loss && dlogits_cls && dlogits_dist, VITK_ERR_NULL);
	VITK_REQUIRE(B > 0 && C > 0 && C < (1ll << 31) && tau > 0.f, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		// class head: hard-label cross entropy weighted 1 - alpha; distillation head: the soft term weighted alpha
		hipLaunchKernelGGL((ce_kd_kernel<T>), dim3((unsigned)vitk_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)logits_cls, labels, (const T*)nullptr, loss,
			(T*)dlogits_cls, B, (int)C, label_smoothing, 1.f - alpha, 0.f, tau);
		VITK_CHECK_LAUNCH();
		hipLaunchKernelGGL((ce_kd_kernel<T>), dim3((unsigned)vitk_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)logits_dist, labels, (const T*)teacher, loss,
			(T*)dlogits_dist, B, (int)C, 0.f, 0.f, alpha, tau);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_drop_cls_rows(const void* x, void* out, int64_t B, int64_t tokens, int64_t d, int dtype, void* stream) {
	VITK_REQUIRE(x && out, VITK_ERR_NULL);
	VITK_REQUIRE(B > 0 && tokens > 1 && d > 0, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((drop_cls_kernel<T>), dim3(vitk_stream_grid(B * (tokens - 1) * d, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)out, B, tokens, d, (int64_t)1);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_mean_pool_fwd(const void* x, void* y, int64_t B, int64_t S, int64_t d, int dtype, void* stream) {
	VITK_REQUIRE(x && y, VITK_ERR_NULL);
	VITK_REQUIRE(B > 0 && S > 0 && d > 0, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((mean_pool_fwd_kernel<T>), dim3(vitk_stream_grid(B * d, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, B, S, d);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_mean_pool_bwd(const void* dy, void* dx, int64_t B, int64_t S, int64_t d, int dtype, void* stream) {
	VITK_REQUIRE(dy && dx, VITK_ERR_NULL);
	VITK_REQUIRE(B > 0 && S > 0 && d > 0, VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((mean_pool_bwd_kernel<T>), dim3(vitk_stream_grid(B * S * d, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, B, S, d);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_ce_kd_loss(const void* logits, const int64_t* labels, const void* teacher, float* loss, void* dlogits, int64_t B, int64_t C,
	float label_smoothing, float alpha, float tau, int dtype, void* stream) {
	VITK_REQUIRE(logits && labels && loss && dlogits, VITK_ERR_NULL);
	VITK_REQUIRE(B > 0 && C > 0 && C < (1 << 24) && tau > 0.f, VITK_ERR_SHAPE);
	if (teacher == nullptr) alpha = 0.f;
	if (alpha == 0.f) teacher = nullptr;
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((ce_kd_kernel<T>), dim3((unsigned)vitk_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)logits, labels, (const T*)teacher, loss,
			(T*)dlogits, B, (int)C, label_smoothing, teacher ? 1.f - alpha : 1.f, alpha, tau);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_adamw(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, void* p_bf16, int64_t n, float lr, float beta1,
	float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
	VITK_REQUIRE(p && g && m && v, VITK_ERR_NULL);
	VITK_REQUIRE(n > 0 && step >= 1, VITK_ERR_SHAPE);
	const double bc1 = 1.0 - pow((double)beta1, (double)step);
	const double bc2 = 1.0 - pow((double)beta2, (double)step);
	hipLaunchKernelGGL(adamw_kernel, dim3(vitk_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, wd_mask, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps,
		weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, (const int64_t*)nullptr, (const float*)nullptr);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

extern "C" int vitk_adamw_device_step(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, void* p_bf16, int64_t n, float lr, float beta1,
	float beta2, float eps, float weight_decay, int64_t* step_state, float grad_scale, void* stream) {
	VITK_REQUIRE(p && g && m && v && step_state, VITK_ERR_NULL);
	VITK_REQUIRE(n > 0, VITK_ERR_SHAPE);
	hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_state, beta1, beta2, (const float*)nullptr);
	VITK_CHECK_LAUNCH();
	hipLaunchKernelGGL(adamw_kernel, dim3(vitk_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, wd_mask, (bf16_t*)p_bf16, n, lr, beta1, beta2, eps,
		weight_decay, 1.f, 1.f, grad_scale, (const int64_t*)step_state, (const float*)nullptr);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

extern "C" int vitk_adamw_device_hparams(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, void* p_bf16, int64_t n, const float* hparams,
	int64_t* step_state, float grad_scale, void* stream) {
	VITK_REQUIRE(p && g && m && v && step_state && hparams, VITK_ERR_NULL);
	VITK_REQUIRE(n > 0, VITK_ERR_SHAPE);
	VITK_REQUIRE(vitk_aligned(hparams, 4), VITK_ERR_ALIGN);
	hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_state, 0.f, 0.f, hparams);
	VITK_CHECK_LAUNCH();
	hipLaunchKernelGGL(adamw_kernel, dim3(vitk_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, wd_mask, (bf16_t*)p_bf16, n, 0.f, 0.f, 0.f, 0.f, 0.f,
		1.f, 1.f, grad_scale, (const int64_t*)step_state, hparams);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

extern "C" int vitk_adamw_advance(int64_t* step_state, const float* hparams, void* stream) {
	VITK_REQUIRE(step_state && hparams, VITK_ERR_NULL);
	VITK_REQUIRE(vitk_aligned(hparams, 4), VITK_ERR_ALIGN);
	hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_state, 0.f, 0.f, hparams);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

extern "C" int vitk_adamw_range(float* p, float* g, float* m, float* v, const uint8_t* wd_mask, void* p_bf16, int64_t n, const float* hparams,
	const int64_t* step_state, float grad_scale, int zero_grad, void* stream) {
	VITK_REQUIRE(p && g && m && v && step_state && hparams, VITK_ERR_NULL);
	VITK_REQUIRE(n > 0, VITK_ERR_SHAPE);
	VITK_REQUIRE(vitk_aligned(hparams, 4), VITK_ERR_ALIGN);
	hipLaunchKernelGGL(adamw_kernel, dim3(vitk_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, (const float*)g, m, v, wd_mask, (bf16_t*)p_bf16, n, 0.f, 0.f, 0.f, 0.f,
		0.f, 1.f, 1.f, grad_scale, step_state, hparams, zero_grad ? g : (float*)nullptr);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}
