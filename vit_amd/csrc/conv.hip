// CvT kernels (all bandwidth-bound, channel-contiguous token-major activations [B, H*W, C]):
//   depthwise k x k convolution of K and V (layers/attention.py:61-62,73-78 of the reference) — operating directly on
//   tokens removes the four 'b (h w) c <-> b c h w' rearrange copies around it;
//   im2col / col2im for the overlapping conv token embedding (layers/embeddings.py:21): the convolution itself is
//   vitk_gemm on the gathered columns, whose [B*Ho*Wo, C_out] output IS the 'b (h w) c' token layout;
//   BatchNorm over token-major activations (layers/embeddings.py:22) with batch statistics, running-stat update and
//   backward.
#include "vitk_common.h"

namespace {

// ---- depthwise conv ----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int64_t B, int H, int W, int C,
	int ks, int stride, int pad, int Ho, int Wo, int64_t ldx) {
	const int64_t n = B * Ho * Wo * C;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int c = (int)(i % C);
		int64_t t = i / C;
		const int wo = (int)(t % Wo);
		t /= Wo;
		const int ho = (int)(t % Ho);
		const int64_t b = t / Ho;
		float acc = 0.f;
		for (int kh = 0; kh < ks; ++kh) {
			const int h = ho * stride - pad + kh;
			if (h < 0 || h >= H) continue;
			for (int kw = 0; kw < ks; ++kw) {
				const int ww = wo * stride - pad + kw;
				if (ww < 0 || ww >= W) continue;
				acc += to_f32<T>(x[((b * H + h) * W + ww) * ldx + c]) * to_f32<T>(w[(c * ks + kh) * ks + kw]);
			}
		}
		y[i] = from_f32<T>(acc);
	}
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ w, T* __restrict__ dx, int64_t B, int H, int W,
	int C, int ks, int stride, int pad, int Ho, int Wo, int64_t lddx) {
	const int64_t n = B * H * W * C;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int c = (int)(i % C);
		int64_t t = i / C;
		const int ww = (int)(t % W);
		t /= W;
		const int h = (int)(t % H);
		const int64_t b = t / H;
		float acc = 0.f;
		for (int kh = 0; kh < ks; ++kh) {
			const int hn = h + pad - kh;
			if (hn < 0 || hn % stride != 0) continue;
			const int ho = hn / stride;
			if (ho >= Ho) continue;
			for (int kw = 0; kw < ks; ++kw) {
				const int wn = ww + pad - kw;
				if (wn < 0 || wn % stride != 0) continue;
				const int wo = wn / stride;
				if (wo >= Wo) continue;
				acc += to_f32<T>(dy[((b * Ho + ho) * Wo + wo) * C + c]) * to_f32<T>(w[(c * ks + kh) * ks + kw]);
			}
		}
		dx[(i / C) * lddx + c] = from_f32<T>(acc);
	}
}

// dw[c, kh, kw] += sum over (b, ho, wo) of dy * x.  Block = 64 channels x DW_TY position lanes; grid.x: channel blocks of
// 64, grid.y: slabs of output positions.  The position lanes of a block are summed through LDS and the block issues
// one float atomic per (channel, tap), so an address sees gridDim.y atomics.
constexpr int DW_TY = 16;
template <typename T>
__global__ __launch_bounds__(64 * DW_TY) void dwconv_bwd_dw_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* dw, int64_t B, int H, int W, int C, int ks,
	int stride, int pad, int Ho, int Wo, int64_t ldx) {
	__shared__ float red[DW_TY][25][64];
	const int tx = threadIdx.x, ty = threadIdx.y;
	const int c = blockIdx.x * 64 + tx;
	const int64_t npos = B * Ho * Wo;
	float acc[25];  // ks <= 5
	for (int j = 0; j < ks * ks; ++j) acc[j] = 0.f;
	if (c < C) {
		for (int64_t pidx = (int64_t)blockIdx.y * DW_TY + ty; pidx < npos; pidx += (int64_t)gridDim.y * DW_TY) {
			const int wo = (int)(pidx % Wo);
			const int ho = (int)((pidx / Wo) % Ho);
			const int64_t b = pidx / ((int64_t)Wo * Ho);
			const float g = to_f32<T>(dy[pidx * C + c]);
			for (int kh = 0; kh < ks; ++kh) {
				const int h = ho * stride - pad + kh;
				if (h < 0 || h >= H) continue;
				for (int kw = 0; kw < ks; ++kw) {
					const int ww = wo * stride - pad + kw;
					if (ww < 0 || ww >= W) continue;
					acc[kh * ks + kw] += g * to_f32<T>(x[((b * H + h) * W + ww) * ldx + c]);
				}
			}
		}
	}
	for (int j = 0; j < ks * ks; ++j) red[ty][j][tx] = acc[j];
	__syncthreads();
	// thread (tx, ty) sums tap j = ty, ty + DW_TY, ... of channel tx over the position lanes
	if (c < C) {
		for (int j = ty; j < ks * ks; j += DW_TY) {
			float v = 0.f;
#pragma unroll
			for (int l = 0; l < DW_TY; ++l) v += red[l][j][tx];
			atomicAdd(&dw[c * ks * ks + j], v);
		}
	}
}

// ---- depthwise 3 x 3 in 16-byte channel groups ----------------------------------------------------------------------
// The kernels above move one 2-byte element per lane and instruction; these own V = 16 bytes of channels per thread
// (C % V == 0, 16-byte aligned tensors), keep the 9 x V taps in registers and walk one image row, so a wave moves 1 KiB
// per instruction and neighbouring outputs find their shared inputs in the vector cache.  Thread = (image, row, channel
// group), channel group fastest.
template <typename T, int V>
__device__ __forceinline__ void load_taps3(const T* __restrict__ w, int c0, float (&wt)[9][V]) {
#pragma unroll
	for (int j = 0; j < V; ++j)
#pragma unroll
		for (int t = 0; t < 9; ++t) wt[t][j] = to_f32<T>(w[(c0 + j) * 9 + t]);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void dwconv3_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w, T* __restrict__ y, int64_t B, int H, int W, int C,
	int stride, int pad, int Ho, int Wo, int64_t ldx) {
	const int ncg = C / V;
	const int64_t n = B * Ho * ncg;
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int cg = (int)(i % ncg);
	const int ho = (int)((i / ncg) % Ho);
	const int64_t b = i / ((int64_t)ncg * Ho);
	const int c0 = cg * V;
	float wt[9][V];
	load_taps3<T, V>(w, c0, wt);
	const T* xb = x + b * H * W * ldx + c0;
	T* yb = y + (b * Ho + ho) * (int64_t)Wo * C + c0;
	for (int wo = 0; wo < Wo; ++wo) {
		float acc[V];
#pragma unroll
		for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
		for (int kh = 0; kh < 3; ++kh) {
			const int h = ho * stride - pad + kh;
			if (h < 0 || h >= H) continue;
#pragma unroll
			for (int kw = 0; kw < 3; ++kw) {
				const int ww = wo * stride - pad + kw;
				if (ww < 0 || ww >= W) continue;
				float v[V];
				load_vec<T, V>(xb + ((int64_t)h * W + ww) * ldx, v);
#pragma unroll
				for (int j = 0; j < V; ++j) acc[j] += v[j] * wt[kh * 3 + kw][j];
			}
		}
		store_vec<T, V>(yb + (int64_t)wo * C, acc);
	}
}

template <typename T, int V>
__global__ __launch_bounds__(256) void dwconv3_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ w, T* __restrict__ dx, int64_t B, int H, int W, int C,
	int stride, int pad, int Ho, int Wo, int64_t lddx) {
	const int ncg = C / V;
	const int64_t n = B * H * ncg;
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int cg = (int)(i % ncg);
	const int h = (int)((i / ncg) % H);
	const int64_t b = i / ((int64_t)ncg * H);
	const int c0 = cg * V;
	float wt[9][V];
	load_taps3<T, V>(w, c0, wt);
	// the output rows this input row feeds, one per tap row (or none)
	int hos[3];
#pragma unroll
	for (int kh = 0; kh < 3; ++kh) {
		const int hn = h + pad - kh;
		hos[kh] = (hn >= 0 && hn % stride == 0 && hn / stride < Ho) ? hn / stride : -1;
	}
	const T* gb = dy + b * Ho * Wo * C + c0;
	T* db = dx + (b * H + h) * (int64_t)W * lddx + c0;
	for (int ww = 0; ww < W; ++ww) {
		float acc[V];
#pragma unroll
		for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
		for (int kw = 0; kw < 3; ++kw) {
			const int wn = ww + pad - kw;
			if (wn < 0 || wn % stride != 0 || wn / stride >= Wo) continue;
			const int wo = wn / stride;
#pragma unroll
			for (int kh = 0; kh < 3; ++kh) {
				if (hos[kh] < 0) continue;
				float v[V];
				load_vec<T, V>(gb + ((int64_t)hos[kh] * Wo + wo) * C, v);
#pragma unroll
				for (int j = 0; j < V; ++j) acc[j] += v[j] * wt[kh * 3 + kw][j];
			}
		}
		store_vec<T, V>(db + (int64_t)ww * lddx, acc);
	}
}

// dw[c, kh, kw] += sum over (b, ho, wo) of dy * x.  Workgroup = 8 channel groups x 32 position lanes; a thread walks the
// output positions lane, lane + 32 * gridDim.y, ... ; the position lanes combine through LDS and the workgroup issues
// one float atomic per (channel, tap).
template <typename T, int V>
__global__ __launch_bounds__(256) void dwconv3_bwd_dw_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* dw, int64_t B, int H, int W, int C, int stride,
	int pad, int Ho, int Wo, int64_t ldx) {
	__shared__ float red[32][8 * V + 1];
	const int ncg = C / V;
	const int cgl = threadIdx.x & 7, pl = threadIdx.x >> 3;
	const int cg = blockIdx.x * 8 + cgl;
	const int c0 = cg * V;
	const int64_t npos = B * Ho * Wo;
	float acc[9][V];
#pragma unroll
	for (int t = 0; t < 9; ++t)
#pragma unroll
		for (int j = 0; j < V; ++j) acc[t][j] = 0.f;
	if (cg < ncg) {
		for (int64_t p = (int64_t)blockIdx.y * 32 + pl; p < npos; p += (int64_t)gridDim.y * 32) {
			const int wo = (int)(p % Wo);
			const int ho = (int)((p / Wo) % Ho);
			const int64_t b = p / ((int64_t)Wo * Ho);
			float g[V];
			load_vec<T, V>(dy + p * C + c0, g);
#pragma unroll
			for (int kh = 0; kh < 3; ++kh) {
				const int h = ho * stride - pad + kh;
				if (h < 0 || h >= H) continue;
#pragma unroll
				for (int kw = 0; kw < 3; ++kw) {
					const int ww = wo * stride - pad + kw;
					if (ww < 0 || ww >= W) continue;
					float v[V];
					load_vec<T, V>(x + ((b * H + h) * W + ww) * ldx + c0, v);
#pragma unroll
					for (int j = 0; j < V; ++j) acc[kh * 3 + kw][j] += g[j] * v[j];
				}
			}
		}
	}
#pragma unroll
	for (int t = 0; t < 9; ++t) {  // (unrolled: a runtime index into acc would move it to scratch memory)
		__syncthreads();
#pragma unroll
		for (int j = 0; j < V; ++j) red[pl][cgl * V + j] = acc[t][j];
		__syncthreads();
		if (threadIdx.x < 8 * V) {
			const int c = blockIdx.x * 8 * V + threadIdx.x;
			if (c < C) {
				float s = 0.f;
#pragma unroll
				for (int l = 0; l < 32; ++l) s += red[l][threadIdx.x];
				atomicAdd(&dw[c * 9 + t], s);
			}
		}
	}
}

// ---- im2col / col2im -----------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ int64_t src_index(int tokens_in, int64_t b, int c, int h, int w, int C, int H, int W) {
	return tokens_in ? ((b * H + h) * W + w) * C + c : ((b * C + c) * H + h) * (int64_t)W + w;
}

template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const T* __restrict__ x, T* __restrict__ cols, int64_t B, int C, int H, int W, int ks, int stride, int pad,
	int Ho, int Wo, int tokens_in, int ld) {
	const int K = C * ks * ks;
	const int64_t n = B * Ho * Wo * ld;  // columns K .. ld - 1 of a row are written as zeros
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int k = (int)(i % ld);
		int64_t t = i / ld;
		if (k >= K) {
			cols[i] = from_f32<T>(0.f);
			continue;
		}
		const int wo = (int)(t % Wo);
		t /= Wo;
		const int ho = (int)(t % Ho);
		const int64_t b = t / Ho;
		const int c = k / (ks * ks), r = k - c * ks * ks;
		const int kh = r / ks, kw = r - kh * ks;
		const int h = ho * stride - pad + kh, w = wo * stride - pad + kw;
		T v = from_f32<T>(0.f);
		if (h >= 0 && h < H && w >= 0 && w < W) v = x[src_index<T>(tokens_in, b, c, h, w, C, H, W)];
		cols[i] = v;
	}
}

template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ cols, T* __restrict__ dx, int64_t B, int C, int H, int W, int ks, int stride, int pad,
	int Ho, int Wo, int tokens_in, int ld) {
	const int K = ld;  // row stride of cols (>= C * ks * ks)
	const int64_t n = B * C * H * W;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		int c, h, w;
		int64_t b;
		if (tokens_in) {
			c = (int)(i % C);
			int64_t t = i / C;
			w = (int)(t % W);
			t /= W;
			h = (int)(t % H);
			b = t / H;
		} else {
			w = (int)(i % W);
			int64_t t = i / W;
			h = (int)(t % H);
			t /= H;
			c = (int)(t % C);
			b = t / C;
		}
		float acc = 0.f;
		for (int kh = 0; kh < ks; ++kh) {
			const int hn = h + pad - kh;
			if (hn < 0 || hn % stride != 0) continue;
			const int ho = hn / stride;
			if (ho >= Ho) continue;
			for (int kw = 0; kw < ks; ++kw) {
				const int wn = w + pad - kw;
				if (wn < 0 || wn % stride != 0) continue;
				const int wo = wn / stride;
				if (wo >= Wo) continue;
				acc += to_f32<T>(cols[((b * Ho + ho) * Wo + wo) * K + (c * ks + kh) * ks + kw]);
			}
		}
		dx[i] = from_f32<T>(acc);
	}
}

// dst[r, 0:cols] = src[r, 0:cols] (converted), dst[r, cols:ld_dst] = 0
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void pad_cols_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int64_t rows, int cols, int64_t ld_src, int ld_dst) {
	const int64_t n = rows * ld_dst;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int c = (int)(i % ld_dst);
		const int64_t r = i / ld_dst;
		dst[i] = c < cols ? from_f32<TO>(to_f32<TI>(src[r * ld_src + c])) : from_f32<TO>(0.f);
	}
}

// ---- BatchNorm over x[rows, C] ------------------------------------------------------------------------------
// pass A: out0[c] += sum_r f0, out1[c] += sum_r f1 with (f0, f1) = (x, -) | ((x-mean)^2, -) | (dy, dy * xhat)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void bn_colreduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mean,
	const float* __restrict__ rstd, float* out0, float* out1, int64_t rows, int C, float* partials) {
	const int c = blockIdx.x * 64 + (threadIdx.x & 63);
	const int sub = threadIdx.x >> 6;
	__shared__ float red0[4][64], red1[4][64];
	float a0 = 0.f, a1 = 0.f;
	if (c < C) {
		const float mu = (MODE >= 1) ? mean[c] : 0.f;
		const float rs = (MODE == 2) ? rstd[c] : 0.f;
		for (int64_t r = (int64_t)blockIdx.y * 4 + sub; r < rows; r += (int64_t)gridDim.y * 4) {
			const float v = to_f32<T>(x[r * C + c]);
			if (MODE == 0) {
				a0 += v;
			} else if (MODE == 1) {
				a0 += (v - mu) * (v - mu);
			} else {
				const float g = to_f32<T>(dy[r * C + c]);
				a0 += g;
				a1 += g * (v - mu) * rs;
			}
		}
	}
	red0[sub][threadIdx.x & 63] = a0;
	red1[sub][threadIdx.x & 63] = a1;
	__syncthreads();
	if (sub == 0 && c < C) {
		const int l = threadIdx.x;
		const float s0 = red0[0][l] + red0[1][l] + red0[2][l] + red0[3][l], s1 = red1[0][l] + red1[1][l] + red1[2][l] + red1[3][l];
		if (partials != nullptr) {  // [row slab][sum][C]: added up in slab order by bn_partials_sum_kernel
			partials[((int64_t)blockIdx.y * 2 + 0) * C + c] = s0;
			if (MODE == 2) partials[((int64_t)blockIdx.y * 2 + 1) * C + c] = s1;
		} else {
			atomicAdd(&out0[c], s0);
			if (MODE == 2) atomicAdd(&out1[c], s1);
		}
	}
}

// out0[c] (and out1[c]) = sum over the row slabs of partials[slab][0 (1)][c], in a FIXED order (64 lanes take the slabs
// l, l + 64, ... in turn; the lane sums are then added 0 + 1 + ... + 63): the column sums of a BatchNorm are then the same
// bits from run to run, which float atomics in arrival order are not (a batch statistic that moves in its last bit flips
// bf16 roundings of the normalised activations, and everything behind them).  Workgroup = 4 channels x 64 lanes (up to
// 2048 slabs: 32 loads per lane).
__global__ __launch_bounds__(256) void bn_partials_sum_kernel(const float* __restrict__ partials, int slabs, int C, float* __restrict__ out0, float* __restrict__ out1) {
	__shared__ float red[2][64][4];
	const int cl = threadIdx.x & 3, lane = threadIdx.x >> 2;
	const int c = blockIdx.x * 4 + cl;
	float a0 = 0.f, a1 = 0.f;
	if (c < C) {
		for (int sl = lane; sl < slabs; sl += 64) {
			a0 += partials[((int64_t)sl * 2 + 0) * C + c];
			if (out1 != nullptr) a1 += partials[((int64_t)sl * 2 + 1) * C + c];
		}
	}
	red[0][lane][cl] = a0;
	red[1][lane][cl] = a1;
	__syncthreads();
	if (lane < 2 && c < C && (lane == 0 || out1 != nullptr)) {  // lane 0: the first sum, lane 1: the second
		float s = 0.f;
		for (int l = 0; l < 64; ++l) s += red[lane][l][cl];
		(lane == 0 ? out0 : out1)[c] = s;
	}
}

// The same in 16-byte channel groups (C % V == 0): workgroup = 16 channel groups x 16 row lanes.
template <typename T, int V, int MODE>
__global__ __launch_bounds__(256) void bn_colreduce_vec_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ mean,
	const float* __restrict__ rstd, float* out0, float* out1, int64_t rows, int C, float* partials, const float* __restrict__ gamma = nullptr,
	const float* __restrict__ beta = nullptr, int act = 0) {
	__shared__ float red0[16][16 * V + 1], red1[16][16 * V + 1];
	const int cgl = threadIdx.x & 15, rl = threadIdx.x >> 4;
	const int cg = blockIdx.x * 16 + cgl;
	const int c0 = cg * V;
	float a0[V], a1[V];
#pragma unroll
	for (int j = 0; j < V; ++j) a0[j] = a1[j] = 0.f;
	if (c0 < C) {
		float mu[V], rs[V], ga[V], be[V];  // (ga, be: with a fused activation, dy is the gradient behind it and the normalised value is formed again)
#pragma unroll
		for (int j = 0; j < V; ++j) {
			mu[j] = (MODE >= 1) ? mean[c0 + j] : 0.f;
			rs[j] = (MODE == 2) ? rstd[c0 + j] : 0.f;
			ga[j] = (MODE == 2 && act != 0) ? gamma[c0 + j] : 0.f;
			be[j] = (MODE == 2 && act != 0) ? beta[c0 + j] : 0.f;
		}
		for (int64_t r = (int64_t)blockIdx.y * 16 + rl; r < rows; r += (int64_t)gridDim.y * 16) {
			float v[V];
			load_vec<T, V>(x + r * C + c0, v);
			if (MODE == 0) {
#pragma unroll
				for (int j = 0; j < V; ++j) a0[j] += v[j];
			} else if (MODE == 1) {
#pragma unroll
				for (int j = 0; j < V; ++j) a0[j] += (v[j] - mu[j]) * (v[j] - mu[j]);
			} else {
				float g[V];
				load_vec<T, V>(dy + r * C + c0, g);
				if (act != 0) {
#pragma unroll
					for (int j = 0; j < V; ++j) g[j] *= act_deriv((v[j] - mu[j]) * rs[j] * ga[j] + be[j], act);
				}
#pragma unroll
				for (int j = 0; j < V; ++j) {
					a0[j] += g[j];
					a1[j] += g[j] * (v[j] - mu[j]) * rs[j];
				}
			}
		}
	}
#pragma unroll
	for (int j = 0; j < V; ++j) {
		red0[rl][cgl * V + j] = a0[j];
		red1[rl][cgl * V + j] = a1[j];
	}
	__syncthreads();
	if (threadIdx.x < 16 * V) {
		const int c = blockIdx.x * 16 * V + threadIdx.x;
		if (c < C) {
			float s0 = 0.f, s1 = 0.f;
#pragma unroll
			for (int l = 0; l < 16; ++l) {
				s0 += red0[l][threadIdx.x];
				s1 += red1[l][threadIdx.x];
			}
			if (partials != nullptr) {
				partials[((int64_t)blockIdx.y * 2 + 0) * C + c] = s0;
				if (MODE == 2) partials[((int64_t)blockIdx.y * 2 + 1) * C + c] = s1;
			} else {
				atomicAdd(&out0[c], s0);
				if (MODE == 2) atomicAdd(&out1[c], s1);
			}
		}
	}
}

// Workgroup = 16 channel groups x 16 row lanes, as in the column reductions: a thread keeps its channels' constants in
// registers and walks rows, so the per-channel parameters are read once per thread instead of once per 16 bytes of data.
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_apply_vec_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
	const float* __restrict__ mean, const float* __restrict__ rstd_or_var, T* __restrict__ y, int64_t rows, int C, float eps, int use_var, int act = 0) {
	const int cg = blockIdx.x * 16 + (threadIdx.x & 15), rl = threadIdx.x >> 4;
	const int c0 = cg * V;
	if (c0 >= C) return;
	float sc[V], sh[V];  // y = x * sc + sh
#pragma unroll
	for (int j = 0; j < V; ++j) {
		const float rs = use_var ? rsqrtf(rstd_or_var[c0 + j] + eps) : rstd_or_var[c0 + j];
		sc[j] = rs * gamma[c0 + j];
		sh[j] = beta[c0 + j] - mean[c0 + j] * sc[j];
	}
	for (int64_t r = (int64_t)blockIdx.y * 16 + rl; r < rows; r += (int64_t)gridDim.y * 16) {
		float v[V], o[V];
		load_vec<T, V>(x + r * C + c0, v);
#pragma unroll
		for (int j = 0; j < V; ++j) o[j] = v[j] * sc[j] + sh[j];
		if (act != 0) {  // the activation behind the normalisation, in the same pass
#pragma unroll
			for (int j = 0; j < V; ++j) o[j] = act_apply(o[j], act);
		}
		store_vec<T, V>(y + r * C + c0, o);
	}
}

// training: dx = A * dy - K * (x - mean) - A * dbeta / n with A = gamma * rstd, K = A * rstd * dgamma / n; eval: dx = A * dy
template <typename T, int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
	const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ dgamma, const float* __restrict__ dbeta, T* __restrict__ dx,
	int64_t rows, int C, float inv_rows, int training, const float* __restrict__ beta = nullptr, int act = 0) {
	const int cg = blockIdx.x * 16 + (threadIdx.x & 15), rl = threadIdx.x >> 4;
	const int c0 = cg * V;
	if (c0 >= C) return;
	float A[V], K[V], D[V], mu[V], be[V];
#pragma unroll
	for (int j = 0; j < V; ++j) {
		A[j] = gamma[c0 + j] * rstd[c0 + j];
		K[j] = training ? A[j] * rstd[c0 + j] * dgamma[c0 + j] * inv_rows : 0.f;
		D[j] = training ? A[j] * dbeta[c0 + j] * inv_rows : 0.f;
		mu[j] = mean[c0 + j];
		be[j] = act != 0 ? beta[c0 + j] : 0.f;
	}
	for (int64_t r = (int64_t)blockIdx.y * 16 + rl; r < rows; r += (int64_t)gridDim.y * 16) {
		float g[V], o[V];
		load_vec<T, V>(dy + r * C + c0, g);
		if (training || act != 0) {
			float v[V];
			load_vec<T, V>(x + r * C + c0, v);
			if (act != 0) {  // dy is the gradient behind the fused activation: times its derivative at the normalised value
#pragma unroll
				for (int j = 0; j < V; ++j) g[j] *= act_deriv((v[j] - mu[j]) * A[j] + be[j], act);
			}
#pragma unroll
			for (int j = 0; j < V; ++j) o[j] = A[j] * g[j] - K[j] * (v[j] - mu[j]) - D[j];
		} else {
#pragma unroll
			for (int j = 0; j < V; ++j) o[j] = A[j] * g[j];
		}
		store_vec<T, V>(dx + r * C + c0, o);
	}
}

// sums -> mean (MODE 0) ; centred sums of squares -> rstd + running-stat update (MODE 1)
template <int MODE>
__global__ void bn_finalize_kernel(float* save_mean, float* save_rstd, float* running_mean, float* running_var, int64_t rows, int C, float eps, float momentum) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= C) return;
	if (MODE == 0) {
		save_mean[c] = save_mean[c] / (float)rows;
	} else {
		const float var = save_rstd[c] / (float)rows;  // biased variance
		save_rstd[c] = rsqrtf(var + eps);
		const float unbiased = rows > 1 ? var * (float)rows / (float)(rows - 1) : var;
		running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * save_mean[c];
		running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
	}
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
	const float* __restrict__ mean, const float* __restrict__ rstd_or_var, T* __restrict__ y, int64_t n, int C, float eps, int use_var) {
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int c = (int)(i % C);
		const float rs = use_var ? rsqrtf(rstd_or_var[c] + eps) : rstd_or_var[c];
		y[i] = from_f32<T>((to_f32<T>(x[i]) - mean[c]) * rs * gamma[c] + beta[c]);
	}
}

// training: dx = gamma * rstd * (dy - dbeta/n - xhat * dgamma/n); eval: dx = gamma * rstd * dy
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
	const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ dgamma, const float* __restrict__ dbeta, T* __restrict__ dx,
	int64_t n, int C, float inv_rows, int training) {
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		const int c = (int)(i % C);
		const float g = to_f32<T>(dy[i]);
		float v = g;
		if (training) {
			const float xhat = (to_f32<T>(x[i]) - mean[c]) * rstd[c];
			v = g - dbeta[c] * inv_rows - xhat * dgamma[c] * inv_rows;
		}
		dx[i] = from_f32<T>(v * gamma[c] * rstd[c]);
	}
}

// eval mode: expose the running statistics in the (mean, rstd) form the backward kernel reads
__global__ void bn_eval_stats_kernel(const float* running_mean, const float* running_var, float* save_mean, float* save_rstd, int C, float eps) {
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c < C) {
		save_mean[c] = running_mean[c];
		save_rstd[c] = rsqrtf(running_var[c] + eps);
	}
}

inline dim3 colgrid(int64_t rows, int C) {
	const unsigned gx = (unsigned)vitk_cdiv(C, 64);
	const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(vitk_cdiv(rows, 16), 1024 / gx));
	return dim3(gx, gy);
}

// grid of the 16-byte-group column reductions: x = blocks of 16 channel groups, y = row slabs (16 row lanes each)
inline dim3 colgrid_vec(int64_t rows, int ngroups) {
	const unsigned gx = (unsigned)vitk_cdiv(ngroups, 16);
	const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(vitk_cdiv(rows, 64), std::max<int64_t>(1, 2048 / gx)));
	return dim3(gx, gy);
}

}  // namespace

#define DISPATCH_T(dtype, ...)             \
	do {                                   \
		if ((dtype) == VITK_F32) {         \
			using T = float;               \
			__VA_ARGS__;                   \
		} else if ((dtype) == VITK_BF16) { \
			using T = bf16_t;              \
			__VA_ARGS__;                   \
		} else                             \
			return VITK_ERR_DTYPE;         \
	} while (0)

static bool conv_geom(int64_t H, int64_t W, int64_t ks, int64_t stride, int64_t pad, int64_t* Ho, int64_t* Wo) {
	if (H <= 0 || W <= 0 || ks <= 0 || ks > 16 || stride <= 0 || pad < 0) return false;
	*Ho = (H + 2 * pad - ks) / stride + 1;
	*Wo = (W + 2 * pad - ks) / stride + 1;
	return *Ho > 0 && *Wo > 0;
}

// ldx / lddx: row stride (elements) of the INPUT-sized tensors x and dx, so that K and V may be column slices of a wider
// buffer (the joint q | k | v projection of CvT's attention); the output-sized tensors y and dy are contiguous [*, C].
extern "C" int vitk_dwconv_fwd_ld(const void* x, int64_t ldx, const void* w, void* y, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ks, int64_t stride,
	int64_t pad, int dtype, void* stream) {
	VITK_REQUIRE(x && w && y, VITK_ERR_NULL);
	int64_t Ho, Wo;
	VITK_REQUIRE(B > 0 && C > 0 && ldx >= C && conv_geom(H, W, ks, stride, pad, &Ho, &Wo), VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		constexpr int V = VecOf<T>::N;
		if (ks == 3 && C % V == 0 && ldx % V == 0 && vitk_aligned(x, 16) && vitk_aligned(y, 16)) {
			hipLaunchKernelGGL((dwconv3_fwd_kernel<T, V>), dim3((unsigned)vitk_cdiv(B * Ho * (C / V), 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)w, (T*)y,
				B, (int)H, (int)W, (int)C, (int)stride, (int)pad, (int)Ho, (int)Wo, ldx);
			VITK_CHECK_LAUNCH();
			return VITK_OK;
		}
		hipLaunchKernelGGL((dwconv_fwd_kernel<T>), dim3(vitk_stream_grid(B * Ho * Wo * C, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)w, (T*)y, B,
			(int)H, (int)W, (int)C, (int)ks, (int)stride, (int)pad, (int)Ho, (int)Wo, ldx);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_dwconv_fwd(const void* x, const void* w, void* y, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ks, int64_t stride, int64_t pad,
	int dtype, void* stream) {
	return vitk_dwconv_fwd_ld(x, C, w, y, B, H, W, C, ks, stride, pad, dtype, stream);
}

extern "C" int vitk_dwconv_bwd_acc(const void* dy, const void* x, int64_t ldx, const void* w, void* dx, int64_t lddx, float* dw, int accumulate_dw, int64_t B,
	int64_t H, int64_t W, int64_t C, int64_t ks, int64_t stride, int64_t pad, int dtype, void* stream) {
	VITK_REQUIRE(dy && x && w && dx && dw, VITK_ERR_NULL);
	int64_t Ho, Wo;
	VITK_REQUIRE(B > 0 && C > 0 && ldx >= C && lddx >= C && conv_geom(H, W, ks, stride, pad, &Ho, &Wo) && ks <= 5, VITK_ERR_SHAPE);
	hipStream_t st = (hipStream_t)stream;
	if (!accumulate_dw) {  // (the kernels add their partial sums into dw)
		hipError_t e = vitk_zero_f32(dw, C * ks * ks, st);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	}
	DISPATCH_T(dtype, {
		constexpr int V = VecOf<T>::N;
		if (ks == 3 && C % V == 0 && ldx % V == 0 && lddx % V == 0 && vitk_aligned(x, 16) && vitk_aligned(dy, 16) && vitk_aligned(dx, 16)) {
			hipLaunchKernelGGL((dwconv3_bwd_dx_kernel<T, V>), dim3((unsigned)vitk_cdiv(B * H * (C / V), 256)), dim3(256), 0, st, (const T*)dy, (const T*)w, (T*)dx, B, (int)H,
				(int)W, (int)C, (int)stride, (int)pad, (int)Ho, (int)Wo, lddx);
			VITK_CHECK_LAUNCH();
			const unsigned gx = (unsigned)vitk_cdiv(C / V, 8);
			const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(vitk_cdiv(B * Ho * Wo, 4 * 32), std::max<int64_t>(1, 1024 / gx)));
			hipLaunchKernelGGL((dwconv3_bwd_dw_kernel<T, V>), dim3(gx, gy), dim3(256), 0, st, (const T*)dy, (const T*)x, dw, B, (int)H, (int)W, (int)C, (int)stride, (int)pad,
				(int)Ho, (int)Wo, ldx);
			VITK_CHECK_LAUNCH();
			return VITK_OK;
		}
		hipLaunchKernelGGL((dwconv_bwd_dx_kernel<T>), dim3(vitk_stream_grid(B * H * W * C, 256)), dim3(256), 0, st, (const T*)dy, (const T*)w, (T*)dx, B, (int)H, (int)W,
			(int)C, (int)ks, (int)stride, (int)pad, (int)Ho, (int)Wo, lddx);
		VITK_CHECK_LAUNCH();
		const unsigned gx = (unsigned)vitk_cdiv(C, 64);
		const unsigned gy = (unsigned)std::max<int64_t>(1, std::min<int64_t>(vitk_cdiv(B * Ho * Wo, 4 * DW_TY), std::max<int64_t>(1, 512 / gx)));
		hipLaunchKernelGGL((dwconv_bwd_dw_kernel<T>), dim3(gx, gy), dim3(64, DW_TY), 0, st, (const T*)dy, (const T*)x, dw, B, (int)H, (int)W, (int)C, (int)ks, (int)stride,
			(int)pad, (int)Ho, (int)Wo, ldx);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_dwconv_bwd_ld(const void* dy, const void* x, int64_t ldx, const void* w, void* dx, int64_t lddx, float* dw, int64_t B, int64_t H, int64_t W,
	int64_t C, int64_t ks, int64_t stride, int64_t pad, int dtype, void* stream) {
	return vitk_dwconv_bwd_acc(dy, x, ldx, w, dx, lddx, dw, 0, B, H, W, C, ks, stride, pad, dtype, stream);
}

extern "C" int vitk_dwconv_bwd(const void* dy, const void* x, const void* w, void* dx, float* dw, int64_t B, int64_t H, int64_t W, int64_t C, int64_t ks,
	int64_t stride, int64_t pad, int dtype, void* stream) {
	return vitk_dwconv_bwd_ld(dy, x, C, w, dx, C, dw, B, H, W, C, ks, stride, pad, dtype, stream);
}

extern "C" int vitk_im2col(const void* x, void* cols, int64_t B, int64_t C, int64_t H, int64_t W, int64_t ks, int64_t stride, int64_t pad, int64_t ld_cols,
	int tokens_in, int dtype, void* stream) {
	VITK_REQUIRE(x && cols, VITK_ERR_NULL);
	int64_t Ho, Wo;
	VITK_REQUIRE(B > 0 && C > 0 && conv_geom(H, W, ks, stride, pad, &Ho, &Wo), VITK_ERR_SHAPE);
	VITK_REQUIRE(ld_cols >= C * ks * ks && ld_cols < (1ll << 31), VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((im2col_kernel<T>), dim3(vitk_stream_grid(B * Ho * Wo * ld_cols, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)cols, B, (int)C,
			(int)H, (int)W, (int)ks, (int)stride, (int)pad, (int)Ho, (int)Wo, tokens_in, (int)ld_cols);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_col2im(const void* cols, void* dx, int64_t B, int64_t C, int64_t H, int64_t W, int64_t ks, int64_t stride, int64_t pad, int64_t ld_cols,
	int tokens_in, int dtype, void* stream) {
	VITK_REQUIRE(cols && dx, VITK_ERR_NULL);
	int64_t Ho, Wo;
	VITK_REQUIRE(B > 0 && C > 0 && conv_geom(H, W, ks, stride, pad, &Ho, &Wo), VITK_ERR_SHAPE);
	VITK_REQUIRE(ld_cols >= C * ks * ks && ld_cols < (1ll << 31), VITK_ERR_SHAPE);
	DISPATCH_T(dtype, {
		hipLaunchKernelGGL((col2im_kernel<T>), dim3(vitk_stream_grid(B * C * H * W, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)cols, (T*)dx, B, (int)C, (int)H,
			(int)W, (int)ks, (int)stride, (int)pad, (int)Ho, (int)Wo, tokens_in, (int)ld_cols);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

// With a workspace (16-byte aligned, at least 2 * slabs * C floats for the grid chosen here: vitk_batchnorm_workspace_bytes)
// the per-slab column sums pass through it and are added in a fixed order: the batch statistics and the parameter gradients
// are then bit-reproducible from run to run.  Without one they are added with float atomics in arrival order.
extern "C" int64_t vitk_batchnorm_workspace_bytes(int64_t rows, int64_t C) {
	if (rows <= 0 || C <= 0) return 0;
	// (the most row slabs any of the paths may choose: element-wise, 16-byte groups of 4 floats or of 8 bf16)
	const unsigned slabs = std::max(colgrid(rows, (int)C).y, std::max(colgrid_vec(rows, (int)vitk_cdiv(C, 4)).y, colgrid_vec(rows, (int)vitk_cdiv(C, 8)).y));
	return (int64_t)slabs * 2 * C * (int64_t)sizeof(float);
}

namespace {
// partial-sum buffer for a reduction launched on `slabs` row slabs, or nullptr when the caller's workspace is absent / too small
inline float* bn_partials(void* ws, int64_t ws_bytes, unsigned slabs, int64_t C) {
	if (ws == nullptr || !vitk_aligned(ws, 16) || ws_bytes < (int64_t)slabs * 2 * C * (int64_t)sizeof(float)) return nullptr;
	return reinterpret_cast<float*>(ws);
}
inline void bn_sum(float* partials, unsigned slabs, int64_t C, float* out0, float* out1, hipStream_t st) {
	if (partials != nullptr) hipLaunchKernelGGL(bn_partials_sum_kernel, dim3((unsigned)vitk_cdiv(C, 4)), dim3(256), 0, st, partials, (int)slabs, (int)C, out0, out1);
}
}  // namespace

// act != 0: y = act(normalised x) in the same pass (16-byte channel groups only: VITK_ERR_UNSUPPORTED otherwise, nothing launched)
static int bn_fwd_impl(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* y, float* save_mean,
	float* save_rstd, int64_t rows, int64_t C, float eps, float momentum, int training, int dtype, int act, void* workspace, int64_t workspace_bytes, void* stream) {
	VITK_REQUIRE(x && gamma && beta && running_mean && running_var && y, VITK_ERR_NULL);
	VITK_REQUIRE(rows > 0 && C > 0 && C < (1 << 30), VITK_ERR_SHAPE);
	VITK_REQUIRE(act >= VITK_ACT_NONE && act <= VITK_ACT_RELU6, VITK_ERR_UNSUPPORTED);
	if (act != 0) {
		const int v = dtype == VITK_BF16 ? 8 : 4;
		if (C % v != 0 || !vitk_aligned(x, 16) || !vitk_aligned(y, 16)) return VITK_ERR_UNSUPPORTED;
	}
	hipStream_t st = (hipStream_t)stream;
	const int64_t n = rows * C;
	if (!training) {
		if (save_mean && save_rstd) hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((unsigned)vitk_cdiv(C, 256)), dim3(256), 0, st, running_mean, running_var, save_mean, save_rstd, (int)C, eps);
		DISPATCH_T(dtype, {
			constexpr int V = VecOf<T>::N;
			if (C % V == 0 && vitk_aligned(x, 16) && vitk_aligned(y, 16))
				hipLaunchKernelGGL((bn_apply_vec_kernel<T, V>), colgrid_vec(rows, (int)C / V), dim3(256), 0, st, (const T*)x, gamma, beta, running_mean, running_var, (T*)y, rows,
					(int)C, eps, 1, act);
			else
				hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(vitk_stream_grid(n, 256)), dim3(256), 0, st, (const T*)x, gamma, beta, running_mean, running_var, (T*)y, n, (int)C,
					eps, 1);
			VITK_CHECK_LAUNCH();
			return VITK_OK;
		});
	}
	VITK_REQUIRE(save_mean && save_rstd, VITK_ERR_NULL);
	hipError_t e = vitk_zero_f32(save_mean, C, st);
	if (e == hipSuccess) e = vitk_zero_f32(save_rstd, C, st);
	if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	const dim3 cg = colgrid(rows, (int)C);
	const unsigned fg = (unsigned)vitk_cdiv(C, 256);
	DISPATCH_T(dtype, {
		constexpr int V = VecOf<T>::N;
		if (C % V == 0 && vitk_aligned(x, 16) && vitk_aligned(y, 16)) {
			const dim3 vg = colgrid_vec(rows, (int)C / V);
			float* part = bn_partials(workspace, workspace_bytes, vg.y, C);
			hipLaunchKernelGGL((bn_colreduce_vec_kernel<T, V, 0>), vg, dim3(256), 0, st, (const T*)x, (const T*)nullptr, nullptr, nullptr, save_mean, nullptr, rows, (int)C, part);
			bn_sum(part, vg.y, C, save_mean, nullptr, st);
			hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(fg), dim3(256), 0, st, save_mean, save_rstd, running_mean, running_var, rows, (int)C, eps, momentum);
			hipLaunchKernelGGL((bn_colreduce_vec_kernel<T, V, 1>), vg, dim3(256), 0, st, (const T*)x, (const T*)nullptr, save_mean, nullptr, save_rstd, nullptr, rows, (int)C, part);
			bn_sum(part, vg.y, C, save_rstd, nullptr, st);
			hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(fg), dim3(256), 0, st, save_mean, save_rstd, running_mean, running_var, rows, (int)C, eps, momentum);
			hipLaunchKernelGGL((bn_apply_vec_kernel<T, V>), vg, dim3(256), 0, st, (const T*)x, gamma, beta, save_mean, save_rstd, (T*)y, rows, (int)C, eps, 0, act);
			VITK_CHECK_LAUNCH();
			return VITK_OK;
		}
		float* part = bn_partials(workspace, workspace_bytes, cg.y, C);
		hipLaunchKernelGGL((bn_colreduce_kernel<T, 0>), cg, dim3(256), 0, st, (const T*)x, (const T*)nullptr, nullptr, nullptr, save_mean, nullptr, rows, (int)C, part);
		bn_sum(part, cg.y, C, save_mean, nullptr, st);
		hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(fg), dim3(256), 0, st, save_mean, save_rstd, running_mean, running_var, rows, (int)C, eps, momentum);
		hipLaunchKernelGGL((bn_colreduce_kernel<T, 1>), cg, dim3(256), 0, st, (const T*)x, (const T*)nullptr, save_mean, nullptr, save_rstd, nullptr, rows, (int)C, part);
		bn_sum(part, cg.y, C, save_rstd, nullptr, st);
		hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(fg), dim3(256), 0, st, save_mean, save_rstd, running_mean, running_var, rows, (int)C, eps, momentum);
		hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(vitk_stream_grid(n, 256)), dim3(256), 0, st, (const T*)x, gamma, beta, save_mean, save_rstd, (T*)y, n, (int)C, eps, 0);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_batchnorm_fwd_ws(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* y, float* save_mean,
	float* save_rstd, int64_t rows, int64_t C, float eps, float momentum, int training, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
	return bn_fwd_impl(x, gamma, beta, running_mean, running_var, y, save_mean, save_rstd, rows, C, eps, momentum, training, dtype, 0, workspace, workspace_bytes, stream);
}

extern "C" int vitk_batchnorm_act_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* y, float* save_mean,
	float* save_rstd, int64_t rows, int64_t C, float eps, float momentum, int training, int act, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
	return bn_fwd_impl(x, gamma, beta, running_mean, running_var, y, save_mean, save_rstd, rows, C, eps, momentum, training, dtype, act, workspace, workspace_bytes, stream);
}

extern "C" int vitk_batchnorm_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, void* y, float* save_mean,
	float* save_rstd, int64_t rows, int64_t C, float eps, float momentum, int training, int dtype, void* stream) {
	return vitk_batchnorm_fwd_ws(x, gamma, beta, running_mean, running_var, y, save_mean, save_rstd, rows, C, eps, momentum, training, dtype, nullptr, 0, stream);
}

// act != 0: dy is the gradient behind the activation that bn_fwd_impl fused; its derivative is taken at the normalised value, formed
// again from x, the statistics, gamma and beta (16-byte channel groups only)
static int bn_bwd_impl(const void* dy, const void* x, const float* gamma, const float* beta, const float* save_mean, const float* save_rstd, void* dx, float* dgamma,
	float* dbeta, int64_t rows, int64_t C, int training, int dtype, int act, void* workspace, int64_t workspace_bytes, void* stream) {
	VITK_REQUIRE(dy && x && gamma && save_mean && save_rstd && dx && dgamma && dbeta, VITK_ERR_NULL);
	VITK_REQUIRE(rows > 0 && C > 0 && C < (1 << 30), VITK_ERR_SHAPE);
	VITK_REQUIRE(act >= VITK_ACT_NONE && act <= VITK_ACT_RELU6, VITK_ERR_UNSUPPORTED);
	if (act != 0) {
		VITK_REQUIRE(beta != nullptr, VITK_ERR_NULL);
		const int v = dtype == VITK_BF16 ? 8 : 4;
		if (C % v != 0 || !vitk_aligned(x, 16) || !vitk_aligned(dy, 16) || !vitk_aligned(dx, 16)) return VITK_ERR_UNSUPPORTED;
	}
	hipStream_t st = (hipStream_t)stream;
	const int64_t n = rows * C;
	hipError_t e = vitk_zero_f32(dgamma, C, st);
	if (e == hipSuccess) e = vitk_zero_f32(dbeta, C, st);
	if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	const dim3 cg = colgrid(rows, (int)C);
	DISPATCH_T(dtype, {
		constexpr int V = VecOf<T>::N;
		if (C % V == 0 && vitk_aligned(x, 16) && vitk_aligned(dy, 16) && vitk_aligned(dx, 16)) {
			const dim3 vg = colgrid_vec(rows, (int)C / V);
			float* part = bn_partials(workspace, workspace_bytes, vg.y, C);
			hipLaunchKernelGGL((bn_colreduce_vec_kernel<T, V, 2>), vg, dim3(256), 0, st, (const T*)x, (const T*)dy, save_mean, save_rstd, dbeta, dgamma, rows, (int)C, part,
				gamma, beta, act);
			bn_sum(part, vg.y, C, dbeta, dgamma, st);
			hipLaunchKernelGGL((bn_bwd_apply_vec_kernel<T, V>), vg, dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, save_mean, save_rstd, dgamma,
				dbeta, (T*)dx, rows, (int)C, 1.0f / (float)rows, training, beta, act);
			VITK_CHECK_LAUNCH();
			return VITK_OK;
		}
		float* part = bn_partials(workspace, workspace_bytes, cg.y, C);
		hipLaunchKernelGGL((bn_colreduce_kernel<T, 2>), cg, dim3(256), 0, st, (const T*)x, (const T*)dy, save_mean, save_rstd, dbeta, dgamma, rows, (int)C, part);
		bn_sum(part, cg.y, C, dbeta, dgamma, st);
		hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(vitk_stream_grid(n, 256)), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, save_mean, save_rstd, dgamma, dbeta,
			(T*)dx, n, (int)C, 1.0f / (float)rows, training);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	});
}

extern "C" int vitk_batchnorm_bwd_ws(const void* dy, const void* x, const float* gamma, const float* save_mean, const float* save_rstd, void* dx, float* dgamma,
	float* dbeta, int64_t rows, int64_t C, int training, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
	return bn_bwd_impl(dy, x, gamma, nullptr, save_mean, save_rstd, dx, dgamma, dbeta, rows, C, training, dtype, 0, workspace, workspace_bytes, stream);
}

extern "C" int vitk_batchnorm_act_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* save_mean, const float* save_rstd, void* dx,
	float* dgamma, float* dbeta, int64_t rows, int64_t C, int training, int act, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
	return bn_bwd_impl(dy, x, gamma, beta, save_mean, save_rstd, dx, dgamma, dbeta, rows, C, training, dtype, act, workspace, workspace_bytes, stream);
}

extern "C" int vitk_batchnorm_bwd(const void* dy, const void* x, const float* gamma, const float* save_mean, const float* save_rstd, void* dx, float* dgamma,
	float* dbeta, int64_t rows, int64_t C, int training, int dtype, void* stream) {
	return vitk_batchnorm_bwd_ws(dy, x, gamma, save_mean, save_rstd, dx, dgamma, dbeta, rows, C, training, dtype, nullptr, 0, stream);
}

extern "C" int vitk_pad_cols(const void* src, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst, int src_dtype, int dst_dtype, void* stream) {
	VITK_REQUIRE(src && dst, VITK_ERR_NULL);
	VITK_REQUIRE(rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= cols && ld_dst < (1ll << 31), VITK_ERR_SHAPE);
	const dim3 grid(vitk_stream_grid(rows * ld_dst, 256));
	hipStream_t st = (hipStream_t)stream;
	if (src_dtype == VITK_F32 && dst_dtype == VITK_F32)
		hipLaunchKernelGGL((pad_cols_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, (float*)dst, rows, (int)cols, ld_src, (int)ld_dst);
	else if (src_dtype == VITK_F32 && dst_dtype == VITK_BF16)
		hipLaunchKernelGGL((pad_cols_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)src, (bf16_t*)dst, rows, (int)cols, ld_src, (int)ld_dst);
	else if (src_dtype == VITK_BF16 && dst_dtype == VITK_BF16)
		hipLaunchKernelGGL((pad_cols_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, rows, (int)cols, ld_src, (int)ld_dst);
	else if (src_dtype == VITK_BF16 && dst_dtype == VITK_F32)
		hipLaunchKernelGGL((pad_cols_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t*)src, (float*)dst, rows, (int)cols, ld_src, (int)ld_dst);
	else
		return VITK_ERR_DTYPE;
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}
