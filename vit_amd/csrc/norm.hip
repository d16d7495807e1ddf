// LayerNorm / RMSNorm forward and backward with optional fused residual add.
//
// Bandwidth-bound: a group of LPR lanes (64 = a whole wavefront; 8, 16 or 32 for rows of at most 1 KB, so that a
// wavefront works on 64 / LPR rows at once and every load instruction still covers whole 128-byte lines) owns one row,
// keeps it in registers (ITERS chunks of VEC elements per lane), and reduces with cross-lane shuffles.  Algorithmic
// HBM bytes per row: forward 2*d*e (+ 2*d*e with the fused residual add), backward 3*d*e (+ d*e with dres).
#include "vitk_common.h"

namespace {

// sum over the LPR consecutive lanes of a row group
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
	for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}

template <typename T, int VEC, int ITERS, bool RMS, int LPR>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
	const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y, T* sum_out,
	float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t rows, int d, float eps, float offset) {
	constexpr int RPW = 64 / LPR;  // rows per wavefront
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const int l = lane % LPR, sub = lane / LPR;
	const int64_t step = (int64_t)gridDim.x * 4 * RPW;
	const int nchunk = d / VEC;
	const float inv_d = 1.0f / (float)d;
	for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * RPW; base < rows; base += step) {
		const int64_t row = base + sub;
		const bool live = row < rows;
		const T* xr = x + row * d;
		float v[ITERS][VEC];
		float s = 0.f;
#pragma unroll
		for (int it = 0; it < ITERS; ++it) {
			const int c = l + it * LPR;
			if (live && c < nchunk) {
				load_vec<T, VEC>(xr + c * VEC, v[it]);
				if (res != nullptr) {
					float r[VEC];
					load_vec<T, VEC>(res + row * d + c * VEC, r);
#pragma unroll
					for (int j = 0; j < VEC; ++j) v[it][j] = round_to<T>(v[it][j] + r[j]);
					store_vec<T, VEC>(sum_out + row * d + c * VEC, v[it]);
				}
#pragma unroll
				for (int j = 0; j < VEC; ++j) s += RMS ? v[it][j] * v[it][j] : v[it][j];
			} else {
#pragma unroll
				for (int j = 0; j < VEC; ++j) v[it][j] = 0.f;
			}
		}
		s = group_sum<LPR>(s);
		float mu = 0.f, rstd;
		if (RMS) {
			rstd = rsqrtf(s * inv_d + eps);
		} else {
			mu = s * inv_d;
			float q = 0.f;
#pragma unroll
			for (int it = 0; it < ITERS; ++it) {
				if (l + it * LPR < nchunk) {
#pragma unroll
					for (int j = 0; j < VEC; ++j) {
						const float t = v[it][j] - mu;
						q += t * t;
					}
				}
			}
			q = group_sum<LPR>(q);
			rstd = rsqrtf(q * inv_d + eps);
		}
		if (l == 0 && live) {
			if (!RMS) mean_out[row] = mu;
			rstd_out[row] = rstd;
		}
#pragma unroll
		for (int it = 0; it < ITERS; ++it) {
			const int c = l + it * LPR;
			if (live && c < nchunk) {
				float g[VEC], o[VEC];
				load_vec<float, VEC>(gamma + c * VEC, g);
				if (RMS) {
#pragma unroll
					for (int j = 0; j < VEC; ++j) o[j] = v[it][j] * rstd * (offset + g[j]);
				} else {
					float b[VEC];
					load_vec<float, VEC>(beta + c * VEC, b);
#pragma unroll
					for (int j = 0; j < VEC; ++j) o[j] = (v[it][j] - mu) * rstd * g[j] + b[j];
				}
				store_vec<T, VEC>(y + row * d + c * VEC, o);
			}
		}
	}
}

// Backward: 8 waves per workgroup (two per SIMD, up to 256 registers each), ONE workgroup per CU.  Measured at 50,432 rows x 768
// (bf16, + dres): 512 threads x 256 workgroups 70.5 us, 640 x 256 74.2, 768 x 256 76.7, 256 x 768 77.8, 384 x 512 85.4,
// 512 x 512 88.3, 1024 x 256 185 (spills).  (More workgroups also mean more rows of column-sum partials to reduce.)  A third
// row of operands in flight per wave (below: two): 89 us at 512 x 256, 105 at 512 x 512, 178 at 768 x 256.
#ifndef VITK_NORM_BWD_THREADS
#define VITK_NORM_BWD_THREADS 512
#endif
#ifndef VITK_NORM_BWD_GRID
#define VITK_NORM_BWD_GRID 256
#endif
constexpr int BWD_THREADS = VITK_NORM_BWD_THREADS;

template <typename T, int VEC, int ITERS, bool RMS, int LPR>
__global__ __launch_bounds__(BWD_THREADS) void norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
	const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd_in,
	const T* __restrict__ dres, T* __restrict__ dx, float* dgamma, float* dbeta, float* partials, int64_t rows, int d, float offset) {
	extern __shared__ __attribute__((aligned(16))) float red[];  // [2][d]
	constexpr int RPW = 64 / LPR;  // rows per wavefront
	const int lane = threadIdx.x & 63;
	const int wave = threadIdx.x >> 6;
	const int l = lane % LPR, sub = lane / LPR;
	constexpr int WAVES = BWD_THREADS / 64;
	const int64_t nwaves = (int64_t)gridDim.x * WAVES * RPW;  // rows between two rows of one lane group
	const int nchunk = d / VEC;
	const float inv_d = 1.0f / (float)d;
	for (int i = threadIdx.x; i < 2 * d; i += BWD_THREADS) red[i] = 0.f;
	__syncthreads();

	float gsc[ITERS][VEC];  // gamma (+offset) for this lane's columns
	float accg[ITERS][VEC], accb[ITERS][VEC];
#pragma unroll
	for (int it = 0; it < ITERS; ++it) {
		const int c = l + it * LPR;
#pragma unroll
		for (int j = 0; j < VEC; ++j) {
			accg[it][j] = 0.f;
			accb[it][j] = 0.f;
			gsc[it][j] = 0.f;
		}
		if (c < nchunk) {
			load_vec<float, VEC>(gamma + c * VEC, gsc[it]);
			if (RMS) {
#pragma unroll
				for (int j = 0; j < VEC; ++j) gsc[it][j] += offset;
			}
		}
	}
	// The operands of the next TWO rows of this wave are in flight (as packed registers) while a row is worked on: the
	// kernel is bound by memory-level parallelism (bytes in flight = bandwidth x latency), not by issue or arithmetic;
	// with one row ahead the 12 waves of a CU keep about 14 MB in flight chip-wide, half of what 8 TB/s needs.
	struct Pre {
		Pack<T, VEC> x[ITERS], dy[ITERS], res[ITERS];
		float mu, rs;
	};
	Pre pa, pb;
	pa.mu = pa.rs = pb.mu = pb.rs = 0.f;
	auto fetch = [&](Pre& p, int64_t row) {
		if (row >= rows) return;
		if (!RMS) p.mu = mean[row];
		p.rs = rstd_in[row];
#pragma unroll
		for (int it = 0; it < ITERS; ++it) {
			const int c = l + it * LPR;
			if (c < nchunk) {
				p.x[it] = *reinterpret_cast<const Pack<T, VEC>*>(x + row * d + c * VEC);
				p.dy[it] = *reinterpret_cast<const Pack<T, VEC>*>(dy + row * d + c * VEC);
				if (dres != nullptr) p.res[it] = *reinterpret_cast<const Pack<T, VEC>*>(dres + row * d + c * VEC);
			}
		}
	};
	// works on row `row`, whose operands are in p, and refills p with row `next`
	auto work = [&](Pre& p, int64_t row, int64_t next) {
		const bool live = row < rows;  // (the loop runs on the wavefront's first row: its other lane groups may be past the end)
		const float mu = p.mu, rs = p.rs;
		float xh[ITERS][VEC], g[ITERS][VEC], rres[ITERS][VEC], dyv[ITERS][VEC];
#pragma unroll
		for (int it = 0; it < ITERS; ++it) {
#pragma unroll
			for (int j = 0; j < VEC; ++j) {
				xh[it][j] = to_f32<T>(p.x[it].v[j]);
				dyv[it][j] = to_f32<T>(p.dy[it].v[j]);
				rres[it][j] = dres != nullptr ? to_f32<T>(p.res[it].v[j]) : 0.f;
			}
		}
		fetch(p, next);
		float s1 = 0.f, s2 = 0.f;
#pragma unroll
		for (int it = 0; it < ITERS; ++it) {
			const int c = l + it * LPR;
			if (live && c < nchunk) {
#pragma unroll
				for (int j = 0; j < VEC; ++j) {
					xh[it][j] = (xh[it][j] - mu) * rs;
					g[it][j] = dyv[it][j] * gsc[it][j];
					s1 += g[it][j];
					s2 += g[it][j] * xh[it][j];
					accg[it][j] += dyv[it][j] * xh[it][j];
					accb[it][j] += dyv[it][j];
				}
			} else {
#pragma unroll
				for (int j = 0; j < VEC; ++j) {
					xh[it][j] = 0.f;
					g[it][j] = 0.f;
				}
			}
		}
		s2 = group_sum<LPR>(s2) * inv_d;
		s1 = RMS ? 0.f : group_sum<LPR>(s1) * inv_d;
#pragma unroll
		for (int it = 0; it < ITERS; ++it) {
			const int c = l + it * LPR;
			if (live && c < nchunk) {
				float o[VEC];
#pragma unroll
				for (int j = 0; j < VEC; ++j) o[j] = (g[it][j] - s1 - xh[it][j] * s2) * rs;
				if (dres != nullptr) {
#pragma unroll
					for (int j = 0; j < VEC; ++j) o[j] += rres[it][j];
				}
				store_vec<T, VEC>(dx + row * d + c * VEC, o);
			}
		}
	};
	const int64_t base0 = ((int64_t)blockIdx.x * WAVES + wave) * RPW;
	fetch(pa, base0 + sub);
	fetch(pb, base0 + sub + nwaves);
	for (int64_t base = base0; base < rows; base += 2 * nwaves) {
		const int64_t row = base + sub;
		work(pa, row, row + 2 * nwaves);
		if (base + nwaves < rows) work(pb, row + nwaves, row + 3 * nwaves);
	}
	// column sums: the lane groups of a wave combine by shuffles, the waves of the block in LDS, then one float atomic
	// per column per block
#pragma unroll
	for (int it = 0; it < ITERS; ++it) {
		const int c = l + it * LPR;
#pragma unroll
		for (int j = 0; j < VEC; ++j) {
#pragma unroll
			for (int o = LPR; o < 64; o <<= 1) {
				accg[it][j] += __shfl_xor(accg[it][j], o, 64);
				if (!RMS) accb[it][j] += __shfl_xor(accb[it][j], o, 64);
			}
		}
		if (sub == 0 && c < nchunk) {
#pragma unroll
			for (int j = 0; j < VEC; ++j) {
				atomicAdd(&red[c * VEC + j], accg[it][j]);
				if (!RMS) atomicAdd(&red[d + c * VEC + j], accb[it][j]);
			}
		}
	}
	__syncthreads();
	if (partials != nullptr) {
		// this workgroup's column sums as one row of [gridDim.x][2 * d]; norm_bwd_reduce_kernel adds the rows up
		const int n = RMS ? d : 2 * d;
		for (int i = threadIdx.x; i < n; i += BWD_THREADS) partials[(int64_t)blockIdx.x * 2 * d + i] = red[i];
		return;
	}
	// no scratch: float atomics.  (Atomics from different XCDs on one address serialise at the memory side, about 40 ns
	// each: 256 workgroups end with a 10 us tail.)  Every workgroup starts at another column, so workgroups that finish
	// together do not queue on the same addresses.
	const int rot = (int)((blockIdx.x * 61u) % (unsigned)d);
	for (int i = threadIdx.x; i < d; i += BWD_THREADS) {
		int c = i + rot;
		if (c >= d) c -= d;
		atomicAdd(&dgamma[c], red[c]);
		if (!RMS) atomicAdd(&dbeta[c], red[d + c]);
	}
}

// dgamma[c] += sum over b of partials[b][c], dbeta[c] += sum over b of partials[b][d + c] (ncol = 2 * d; RMSNorm: ncol = d):
// 32 columns x 32 row lanes per workgroup (a row of 32 floats is one 128-byte line; at most 8 independent loads per
// thread for 256 partial rows), the 32 partial sums of a column combined in LDS.  One writer per column.
__global__ __launch_bounds__(1024) void norm_bwd_reduce_kernel(const float* __restrict__ partials, int nblocks, int d, int ncol, float* dgamma, float* dbeta) {
	__shared__ float sm[32][33];
	const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
	const int c = blockIdx.x * 32 + cx;
	float s = 0.f;
	if (c < ncol) {
#pragma unroll 8
		for (int b = ry; b < nblocks; b += 32) s += partials[(int64_t)b * 2 * d + c];
	}
	sm[ry][cx] = s;
	__syncthreads();
	if (ry == 0 && c < ncol) {
#pragma unroll
		for (int r = 1; r < 32; ++r) s += sm[r][cx];
		if (c < d) dgamma[c] += s;
		else dbeta[c - d] += s;
	}
}

// ---- DyT (layers/norm.py:35-45): y = gamma * tanh(alpha * x) + beta, alpha a learned scalar ----------------------------
// Elementwise and bandwidth-bound.  Thread t of a workgroup owns the column chunks t, t + 256, ... (at most DYT_CH of
// VEC columns each) and walks the rows of its workgroup's slab, so gamma / beta and the column sums of the backward
// stay in registers; a workgroup ends with one float atomic per column (and one for alpha).
constexpr int DYT_CH = 4;

template <typename T, int VEC>
__global__ __launch_bounds__(256) void dyt_fwd_kernel(const T* __restrict__ x, const float* __restrict__ alpha, const float* __restrict__ gamma,
	const float* __restrict__ beta, T* __restrict__ y, int64_t rows, int d) {
	const int nchunk = d / VEC;
	const float al = alpha[0];
	float ga[DYT_CH][VEC], be[DYT_CH][VEC];
#pragma unroll
	for (int k = 0; k < DYT_CH; ++k) {
		const int c = threadIdx.x + 256 * k;
		if (c < nchunk) {
			load_vec<float, VEC>(gamma + c * VEC, ga[k]);
			load_vec<float, VEC>(beta + c * VEC, be[k]);
		}
	}
	for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
#pragma unroll
		for (int k = 0; k < DYT_CH; ++k) {
			const int c = threadIdx.x + 256 * k;
			if (c < nchunk) {
				float v[VEC], o[VEC];
				load_vec<T, VEC>(x + row * d + c * VEC, v);
#pragma unroll
				for (int j = 0; j < VEC; ++j) o[j] = ga[k][j] * tanhf(al * v[j]) + be[k][j];
				store_vec<T, VEC>(y + row * d + c * VEC, o);
			}
		}
	}
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void dyt_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ alpha,
	const float* __restrict__ gamma, T* __restrict__ dx, float* dalpha, float* dgamma, float* dbeta, int64_t rows, int d) {
	__shared__ float red[4];
	const int nchunk = d / VEC;
	const float al = alpha[0];
	float ga[DYT_CH][VEC], accg[DYT_CH][VEC], accb[DYT_CH][VEC];
	float acca = 0.f;
#pragma unroll
	for (int k = 0; k < DYT_CH; ++k) {
		const int c = threadIdx.x + 256 * k;
#pragma unroll
		for (int j = 0; j < VEC; ++j) accg[k][j] = accb[k][j] = ga[k][j] = 0.f;
		if (c < nchunk) load_vec<float, VEC>(gamma + c * VEC, ga[k]);
	}
	for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
#pragma unroll
		for (int k = 0; k < DYT_CH; ++k) {
			const int c = threadIdx.x + 256 * k;
			if (c < nchunk) {
				float v[VEC], g[VEC], o[VEC];
				load_vec<T, VEC>(x + row * d + c * VEC, v);
				load_vec<T, VEC>(dy + row * d + c * VEC, g);
#pragma unroll
				for (int j = 0; j < VEC; ++j) {
					const float t = tanhf(al * v[j]);
					const float u = g[j] * ga[k][j] * (1.f - t * t);  // d loss / d (alpha x)
					accg[k][j] += g[j] * t;
					accb[k][j] += g[j];
					acca += u * v[j];
					o[j] = u * al;
				}
				store_vec<T, VEC>(dx + row * d + c * VEC, o);
			}
		}
	}
#pragma unroll
	for (int k = 0; k < DYT_CH; ++k) {
		const int c = threadIdx.x + 256 * k;
		if (c < nchunk) {
#pragma unroll
			for (int j = 0; j < VEC; ++j) {
				atomicAdd(&dgamma[c * VEC + j], accg[k][j]);
				atomicAdd(&dbeta[c * VEC + j], accb[k][j]);
			}
		}
	}
	acca = wave_sum(acca);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acca;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd(dalpha, red[0] + red[1] + red[2] + red[3]);
}

template <typename T, int VEC>
int launch_dyt(const void* dy, const void* x, const float* alpha, const float* gamma, const float* beta, void* out, float* dalpha, float* dgamma, float* dbeta,
	int64_t rows, int d, hipStream_t st) {
	const int grid = (int)std::min<int64_t>(rows, 1024);
	if (dy == nullptr)
		hipLaunchKernelGGL((dyt_fwd_kernel<T, VEC>), dim3(grid), dim3(256), 0, st, (const T*)x, alpha, gamma, beta, (T*)out, rows, d);
	else
		hipLaunchKernelGGL((dyt_bwd_kernel<T, VEC>), dim3(grid), dim3(256), 0, st, (const T*)dy, (const T*)x, alpha, gamma, (T*)out, dalpha, dgamma, dbeta, rows, d);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

int run_dyt(const void* dy, const void* x, const float* alpha, const float* gamma, const float* beta, void* out, float* dalpha, float* dgamma, float* dbeta,
	int64_t rows, int64_t d, int dtype, hipStream_t st) {
	VITK_REQUIRE(rows > 0 && d > 0, VITK_ERR_SHAPE);
	VITK_REQUIRE(dtype == VITK_F32 || dtype == VITK_BF16, VITK_ERR_DTYPE);
	const bool vec = d % 4 == 0 && vitk_aligned(x, 16) && vitk_aligned(out, 16) && (dy == nullptr || vitk_aligned(dy, 16)) && vitk_aligned(gamma, 16) &&
		(beta == nullptr || vitk_aligned(beta, 16));
	VITK_REQUIRE(d <= (vec ? 4 : 1) * 256 * DYT_CH, VITK_ERR_UNSUPPORTED);
	if (dtype == VITK_F32)
		return vec ? launch_dyt<float, 4>(dy, x, alpha, gamma, beta, out, dalpha, dgamma, dbeta, rows, (int)d, st)
				   : launch_dyt<float, 1>(dy, x, alpha, gamma, beta, out, dalpha, dgamma, dbeta, rows, (int)d, st);
	return vec ? launch_dyt<bf16_t, 4>(dy, x, alpha, gamma, beta, out, dalpha, dgamma, dbeta, rows, (int)d, st)
			   : launch_dyt<bf16_t, 1>(dy, x, alpha, gamma, beta, out, dalpha, dgamma, dbeta, rows, (int)d, st);
}

struct NormArgs {
	const void *x, *res, *dy, *dres;
	const float *gamma, *beta, *mean_in, *rstd_in;
	void *y, *sum_out, *dx;
	float *mean, *rstd, *dgamma, *dbeta;
	void* workspace;  // backward: room for the per-workgroup column sums, or null
	int64_t workspace_bytes;
	int64_t rows;
	int d;
	float eps, offset;
};

template <typename T, int VEC, int ITERS, bool RMS, int LPR = 64>
int launch_fwd(const NormArgs& a, hipStream_t st) {
	int grid = (int)std::min<int64_t>(vitk_cdiv(a.rows, 4 * (64 / LPR)), 8192);
	hipLaunchKernelGGL((norm_fwd_kernel<T, VEC, ITERS, RMS, LPR>), dim3(grid), dim3(256), 0, st, (const T*)a.x, (const T*)a.res, a.gamma, a.beta,
		(T*)a.y, (T*)a.sum_out, a.mean, a.rstd, a.rows, a.d, a.eps, a.offset);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}
template <typename T, int VEC, int ITERS, bool RMS, int LPR = 64>
int launch_bwd(const NormArgs& a, hipStream_t st) {
	int grid = (int)std::min<int64_t>(vitk_cdiv(a.rows, (BWD_THREADS / 64) * (64 / LPR)), VITK_NORM_BWD_GRID);
	size_t lds = (size_t)2 * a.d * sizeof(float);
	float* partials = nullptr;
	if (a.workspace != nullptr && vitk_aligned(a.workspace, 16) && a.workspace_bytes >= (int64_t)grid * 2 * a.d * (int64_t)sizeof(float))
		partials = reinterpret_cast<float*>(a.workspace);
	hipLaunchKernelGGL((norm_bwd_kernel<T, VEC, ITERS, RMS, LPR>), dim3(grid), dim3(BWD_THREADS), lds, st, (const T*)a.dy, (const T*)a.x, a.gamma, a.mean_in,
		a.rstd_in, (const T*)a.dres, (T*)a.dx, a.dgamma, a.dbeta, partials, a.rows, a.d, a.offset);
	VITK_CHECK_LAUNCH();
	if (partials != nullptr) {
		const int ncol = RMS ? a.d : 2 * a.d;
		hipLaunchKernelGGL(norm_bwd_reduce_kernel, dim3((unsigned)vitk_cdiv(ncol, 32)), dim3(1024), 0, st, partials, grid, a.d, ncol, a.dgamma, a.dbeta);
		VITK_CHECK_LAUNCH();
	}
	return VITK_OK;
}

// Rows of at most 1 KB: several rows per wavefront.  The smallest lane group (8, 16 or 32 lanes) that holds the row in at
// most three chunks per lane; returns 1 (not a status code) when the row does not fit this form.
template <typename T, bool RMS, bool FWD, int VEC>
int launch_small_rows(const NormArgs& a, hipStream_t st) {
	const int nchunk = a.d / VEC;
	const int lpr = nchunk <= 24 ? 8 : nchunk <= 48 ? 16 : nchunk <= 96 ? 32 : 0;
	if (a.d % VEC != 0 || lpr == 0) return 1;
	const int it = (int)vitk_cdiv(nchunk, lpr);
#define NORM_SMALL(LPR_, IT_) \
	if (lpr == LPR_ && it == IT_) { \
		if constexpr (FWD) return launch_fwd<T, VEC, IT_, RMS, LPR_>(a, st); \
		else return launch_bwd<T, VEC, IT_, RMS, LPR_>(a, st); \
	}
	NORM_SMALL(8, 1) NORM_SMALL(8, 2) NORM_SMALL(8, 3) NORM_SMALL(16, 2) NORM_SMALL(16, 3) NORM_SMALL(32, 2) NORM_SMALL(32, 3)
#undef NORM_SMALL
	return 1;
}

template <typename T, bool RMS, bool FWD>
int dispatch(const NormArgs& a, bool vec_ok, hipStream_t st) {
	constexpr int V = VecOf<T>::N;
#define NORM_GO(VEC_, IT_) return FWD ? launch_fwd<T, VEC_, IT_, RMS>(a, st) : launch_bwd<T, VEC_, IT_, RMS>(a, st)
	static const bool small_rows = getenv("VITK_NO_NORM_SMALL_ROWS") == nullptr;  // A-B switch
	if (small_rows && vec_ok && (size_t)a.d * sizeof(T) <= 1024) {
		// (the backward is register-heavy: half-width chunks, as below)
		const int rc = launch_small_rows<T, RMS, FWD, FWD ? V : V / 2>(a, st);
		if (rc != 1) return rc;
	}
	if (!FWD && vec_ok && a.d % (V / 2) == 0) {
		// backward (register-heavy): if half-width accesses leave fewer lane slots idle (d = 768: 3 x 64 x 4 exactly,
		// against 2 x 64 x 8 with a quarter idle), the smaller footprint keeps 4 waves per SIMD resident
		constexpr int H = V / 2;
		const int it_full = a.d % V == 0 ? (int)vitk_cdiv(a.d / V, 64) : 1 << 20;
		const int it_half = (int)vitk_cdiv(a.d / H, 64);
		if (it_half <= 3 && it_half * H < it_full * V) {
			if (it_half <= 1) return launch_bwd<T, H, 1, RMS>(a, st);
			if (it_half <= 2) return launch_bwd<T, H, 2, RMS>(a, st);
			return launch_bwd<T, H, 3, RMS>(a, st);
		}
	}
	if (vec_ok && a.d % V == 0) {
		const int iters = (int)vitk_cdiv(a.d / V, 64);
		if (iters <= 1) NORM_GO(V, 1);
		if (iters <= 2) NORM_GO(V, 2);
		if (iters <= 4) NORM_GO(V, 4);
		if (iters <= 8) NORM_GO(V, 8);
		if (iters <= 16) NORM_GO(V, 16);
		return VITK_ERR_UNSUPPORTED;
	}
	// rows that cannot be read in 16-byte pieces (d or a pointer unaligned): one element per lane and step, d <= 1024
	const int iters = (int)vitk_cdiv(a.d, 64);
	if (iters <= 1) NORM_GO(1, 1);
	if (iters <= 4) NORM_GO(1, 4);
	if (iters <= 16) NORM_GO(1, 16);
	return VITK_ERR_UNSUPPORTED;
#undef NORM_GO
}

template <bool RMS, bool FWD>
int run(const NormArgs& a, int dtype, hipStream_t st) {
	VITK_REQUIRE(a.rows > 0 && a.d > 0, VITK_ERR_SHAPE);
	VITK_REQUIRE(a.d <= 4096, VITK_ERR_UNSUPPORTED);  // the row lives in registers
	bool vec_ok = vitk_aligned(a.x, 16) && vitk_aligned(a.gamma, 16) && (a.beta == nullptr || vitk_aligned(a.beta, 16)) &&
		(a.res == nullptr || vitk_aligned(a.res, 16)) && (a.y == nullptr || vitk_aligned(a.y, 16)) && (a.sum_out == nullptr || vitk_aligned(a.sum_out, 16)) &&
		(a.dy == nullptr || vitk_aligned(a.dy, 16)) && (a.dres == nullptr || vitk_aligned(a.dres, 16)) && (a.dx == nullptr || vitk_aligned(a.dx, 16));
	if (dtype == VITK_F32) return dispatch<float, RMS, FWD>(a, vec_ok, st);
	if (dtype == VITK_BF16) return dispatch<bf16_t, RMS, FWD>(a, vec_ok, st);
	return VITK_ERR_DTYPE;
}

}  // namespace

extern "C" int vitk_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* y, void* sum_out, float* mean,
	float* rstd, int64_t rows, int64_t d, float eps, int dtype, void* stream) {
	VITK_REQUIRE(x && gamma && beta && y && mean && rstd, VITK_ERR_NULL);
	VITK_REQUIRE(res == nullptr || sum_out != nullptr, VITK_ERR_NULL);
	NormArgs a{};
	a.x = x; a.res = res; a.gamma = gamma; a.beta = beta; a.y = y; a.sum_out = sum_out; a.mean = mean; a.rstd = rstd;
	a.rows = rows; a.d = (int)d; a.eps = eps; a.offset = 0.f;
	return run<false, true>(a, dtype, (hipStream_t)stream);
}

extern "C" int vitk_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
	void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t d, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
	VITK_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, VITK_ERR_NULL);
	NormArgs a{};
	a.workspace = workspace; a.workspace_bytes = workspace_bytes;
	a.dy = dy; a.x = x; a.gamma = gamma; a.mean_in = mean; a.rstd_in = rstd; a.dres = dres; a.dx = dx; a.dgamma = dgamma; a.dbeta = dbeta;
	a.rows = rows; a.d = (int)d; a.offset = 0.f;
	return run<false, false>(a, dtype, (hipStream_t)stream);
}

extern "C" int vitk_rmsnorm_fwd(const void* x, const void* res, const float* weight, void* y, void* sum_out, float* rstd, int64_t rows,
	int64_t d, float eps, int unit_offset, int dtype, void* stream) {
	VITK_REQUIRE(x && weight && y && rstd, VITK_ERR_NULL);
	VITK_REQUIRE(res == nullptr || sum_out != nullptr, VITK_ERR_NULL);
	NormArgs a{};
	a.x = x; a.res = res; a.gamma = weight; a.y = y; a.sum_out = sum_out; a.rstd = rstd;
	a.rows = rows; a.d = (int)d; a.eps = eps; a.offset = unit_offset ? 1.f : 0.f;
	return run<true, true>(a, dtype, (hipStream_t)stream);
}

extern "C" int vitk_rmsnorm_bwd(const void* dy, const void* x, const float* weight, const float* rstd, const void* dres, void* dx,
	float* dweight, int64_t rows, int64_t d, int unit_offset, int dtype, void* workspace, int64_t workspace_bytes, void* stream) {
	VITK_REQUIRE(dy && x && weight && rstd && dx && dweight, VITK_ERR_NULL);
	NormArgs a{};
	a.workspace = workspace; a.workspace_bytes = workspace_bytes;
	a.dy = dy; a.x = x; a.gamma = weight; a.rstd_in = rstd; a.dres = dres; a.dx = dx; a.dgamma = dweight;
	a.rows = rows; a.d = (int)d; a.offset = unit_offset ? 1.f : 0.f;
	return run<true, false>(a, dtype, (hipStream_t)stream);
}

extern "C" int vitk_dyt_fwd(const void* x, const float* alpha, const float* gamma, const float* beta, void* y, int64_t rows, int64_t d, int dtype, void* stream) {
	VITK_REQUIRE(x && alpha && gamma && beta && y, VITK_ERR_NULL);
	return run_dyt(nullptr, x, alpha, gamma, beta, y, nullptr, nullptr, nullptr, rows, d, dtype, (hipStream_t)stream);
}

extern "C" int vitk_dyt_bwd(const void* dy, const void* x, const float* alpha, const float* gamma, void* dx, float* dalpha, float* dgamma, float* dbeta,
	int64_t rows, int64_t d, int dtype, void* stream) {
	VITK_REQUIRE(dy && x && alpha && gamma && dx && dalpha && dgamma && dbeta, VITK_ERR_NULL);
	return run_dyt(dy, x, alpha, gamma, nullptr, dx, dalpha, dgamma, dbeta, rows, d, dtype, (hipStream_t)stream);
}
