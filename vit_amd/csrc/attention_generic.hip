// Generic scaled-dot-product attention (forward and backward) for any head size <= 64, any query /
// key counts (Sq != Skv allowed) and float32 or bfloat16 storage, float32 arithmetic on the vector ALU.
// One lane owns one query row (forward, dQ) or one key row (dK, dV); the other side is streamed
// through LDS in 32-row tiles and broadcast to all lanes.  Used for float32 parity runs and for
// shapes the bf16 MFMA kernel does not cover (head sizes 12 / 24, very long key sequences).
#include "attention.h"

namespace {

constexpr int TK = 32;  // rows of the streamed side per LDS tile

// cooperative load of rows [r0, r0+TK) x [0, DH) of a [rows_total, ld] matrix (column offset applied by caller)
template <typename T, int DH>
__device__ __forceinline__ void stage_rows(float (*dst)[DH], const T* src, int64_t ld, int r0, int rows_total, int dh) {
	for (int idx = threadIdx.x; idx < TK * DH; idx += 64) {
		const int j = idx / DH, d = idx - j * DH;
		const int r = r0 + j;
		dst[j][d] = (r < rows_total && d < dh) ? to_f32<T>(src[(int64_t)r * ld + d]) : 0.f;
	}
}

template <typename T, int DH>
__global__ __launch_bounds__(64) void attn_fwd_generic_kernel(AttnArgs a) {
	attn_enter(a);
	__shared__ float Ks[TK][DH];
	__shared__ float Vs[TK][DH];
	const int b = blockIdx.z, h = blockIdx.y;
	const int qi = blockIdx.x * 64 + threadIdx.x;
	const bool valid = qi < a.Sq;
	const int dh = a.dh;
	const T* q = reinterpret_cast<const T*>(a.q) + ((int64_t)b * a.Sq + (valid ? qi : 0)) * a.ldq + h * dh;
	const T* kb = reinterpret_cast<const T*>(a.k) + (int64_t)b * a.Skv * a.ldk + h * dh;
	const T* vb = reinterpret_cast<const T*>(a.v) + (int64_t)b * a.Skv * a.ldv + h * dh;
	float qr[DH], o[DH];
#pragma unroll
	for (int d = 0; d < DH; ++d) {
		qr[d] = (d < dh) ? to_f32<T>(q[d < dh ? d : 0]) * a.scale : 0.f;
		o[d] = 0.f;
	}
	float m = -INFINITY, l = 0.f;
	const bool drop = a.drop_thr16 != 0;
	const uint32_t rk = attn_row_key(attn_head_key(a, b * a.n_head + h), qi);
	for (int k0 = 0; k0 < a.Skv; k0 += TK) {
		__syncthreads();
		stage_rows<T, DH>(Ks, kb, a.ldk, k0, a.Skv, dh);
		stage_rows<T, DH>(Vs, vb, a.ldv, k0, a.Skv, dh);
		__syncthreads();
		const int nk = min(TK, a.Skv - k0);
		float s[TK];
		float tmax = -INFINITY;
#pragma unroll
		for (int j = 0; j < TK; ++j) {
			float acc = 0.f;
#pragma unroll
			for (int d = 0; d < DH; ++d) acc += qr[d] * Ks[j][d];
			s[j] = (j < nk) ? acc : -INFINITY;
			tmax = fmaxf(tmax, s[j]);
		}
		const float mnew = fmaxf(m, tmax);
		const float alpha = __expf(m - mnew);
		l *= alpha;
#pragma unroll
		for (int d = 0; d < DH; ++d) o[d] *= alpha;
#pragma unroll
		for (int j = 0; j < TK; ++j) {
			const float p = __expf(s[j] - mnew);  // exp(-inf) = 0 for masked keys
			l += p;
			// dropout leaves the softmax normaliser alone and removes / rescales terms of the P V sum
			const float pv = drop ? p * attn_keep(attn_pair_bits(rk, (k0 + j) >> 1), k0 + j, a.drop_thr16, a.drop_scale) : p;
#pragma unroll
			for (int d = 0; d < DH; ++d) o[d] += pv * Vs[j][d];
		}
		m = mnew;
	}
	if (valid) {
		const float inv = 1.f / l;
		T* op = reinterpret_cast<T*>(a.out_w) + ((int64_t)b * a.Sq + qi) * a.ldo + h * dh;
#pragma unroll
		for (int d = 0; d < DH; ++d)
			if (d < dh) op[d] = from_f32<T>(o[d] * inv);
		a.lse[((int64_t)b * a.n_head + h) * a.Sq + qi] = m + __logf(l);
	}
}

// dQ: lane = query.  q and dO rows live in LDS with a padded stride so each lane reads its own row conflict-free.
template <typename T, int DH>
__global__ __launch_bounds__(64) void attn_bwd_dq_kernel(AttnArgs a) {
	attn_enter(a);
	__shared__ float Ks[TK][DH];
	__shared__ float Vs[TK][DH];
	__shared__ float Qs[64][DH + 1];
	__shared__ float Gs[64][DH + 1];
	const int b = blockIdx.z, h = blockIdx.y;
	const int lane = threadIdx.x;
	const int q0 = blockIdx.x * 64;
	const int qi = q0 + lane;
	const bool valid = qi < a.Sq;
	const int dh = a.dh;
	const T* qb = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.Sq * a.ldq + h * dh;
	const T* gb = reinterpret_cast<const T*>(a.dout) + (int64_t)b * a.Sq * a.ldo + h * dh;
	const T* ob = reinterpret_cast<const T*>(a.out) + (int64_t)b * a.Sq * a.ldo + h * dh;
	const T* kb = reinterpret_cast<const T*>(a.k) + (int64_t)b * a.Skv * a.ldk + h * dh;
	const T* vb = reinterpret_cast<const T*>(a.v) + (int64_t)b * a.Skv * a.ldv + h * dh;
	for (int idx = lane; idx < 64 * DH; idx += 64) {
		const int j = idx / DH, d = idx - j * DH;
		const bool ok = (q0 + j < a.Sq) && d < dh;
		Qs[j][d] = ok ? to_f32<T>(qb[(int64_t)(q0 + j) * a.ldq + d]) : 0.f;
		Gs[j][d] = ok ? to_f32<T>(gb[(int64_t)(q0 + j) * a.ldo + d]) : 0.f;
	}
	float delta = 0.f;
	if (valid)
		for (int d = 0; d < dh; ++d) delta += to_f32<T>(gb[(int64_t)qi * a.ldo + d]) * to_f32<T>(ob[(int64_t)qi * a.ldo + d]);
	const float lse = valid ? a.lse[((int64_t)b * a.n_head + h) * a.Sq + qi] : 0.f;
	const bool drop = a.drop_thr16 != 0;
	const uint32_t rk = attn_row_key(attn_head_key(a, b * a.n_head + h), qi);
	float dq[DH];
#pragma unroll
	for (int d = 0; d < DH; ++d) dq[d] = 0.f;
	for (int k0 = 0; k0 < a.Skv; k0 += TK) {
		__syncthreads();
		stage_rows<T, DH>(Ks, kb, a.ldk, k0, a.Skv, dh);
		stage_rows<T, DH>(Vs, vb, a.ldv, k0, a.Skv, dh);
		__syncthreads();
		const int nk = min(TK, a.Skv - k0);
		for (int j = 0; j < nk; ++j) {
			float s = 0.f, dp = 0.f;
#pragma unroll
			for (int d = 0; d < DH; ++d) {
				s += Qs[lane][d] * Ks[j][d];
				dp += Gs[lane][d] * Vs[j][d];
			}
			const float p = __expf(s * a.scale - lse);
			if (drop) dp *= attn_keep(attn_pair_bits(rk, (k0 + j) >> 1), k0 + j, a.drop_thr16, a.drop_scale);
			const float ds = p * (dp - delta) * a.scale;
#pragma unroll
			for (int d = 0; d < DH; ++d) dq[d] += ds * Ks[j][d];
		}
	}
	if (valid) {
		T* dqp = reinterpret_cast<T*>(a.dq) + ((int64_t)b * a.Sq + qi) * a.lddq + h * dh;
#pragma unroll
		for (int d = 0; d < DH; ++d)
			if (d < dh) dqp[d] = from_f32<T>(dq[d]);
	}
}

// dK, dV: lane = key.  Q, dO, O are streamed in 32-query tiles.
template <typename T, int DH>
__global__ __launch_bounds__(64) void attn_bwd_dkv_kernel(AttnArgs a) {
	attn_enter(a);
	__shared__ float Qs[TK][DH];
	__shared__ float Gs[TK][DH];
	__shared__ float Ks[64][DH + 1];
	__shared__ float Vs[64][DH + 1];
	__shared__ float Ls[TK], Ds[TK];
	__shared__ uint32_t Rk[TK];
	const bool drop = a.drop_thr16 != 0;
	const int b = blockIdx.z, h = blockIdx.y;
	const int lane = threadIdx.x;
	const int kbase = blockIdx.x * 64;
	const int ki = kbase + lane;
	const bool valid = ki < a.Skv;
	const int dh = a.dh;
	const T* qb = reinterpret_cast<const T*>(a.q) + (int64_t)b * a.Sq * a.ldq + h * dh;
	const T* gb = reinterpret_cast<const T*>(a.dout) + (int64_t)b * a.Sq * a.ldo + h * dh;
	const T* ob = reinterpret_cast<const T*>(a.out) + (int64_t)b * a.Sq * a.ldo + h * dh;
	const T* kb = reinterpret_cast<const T*>(a.k) + (int64_t)b * a.Skv * a.ldk + h * dh;
	const T* vb = reinterpret_cast<const T*>(a.v) + (int64_t)b * a.Skv * a.ldv + h * dh;
	for (int idx = lane; idx < 64 * DH; idx += 64) {
		const int j = idx / DH, d = idx - j * DH;
		const bool ok = (kbase + j < a.Skv) && d < dh;
		Ks[j][d] = ok ? to_f32<T>(kb[(int64_t)(kbase + j) * a.ldk + d]) : 0.f;
		Vs[j][d] = ok ? to_f32<T>(vb[(int64_t)(kbase + j) * a.ldv + d]) : 0.f;
	}
	float dk[DH], dv[DH];
#pragma unroll
	for (int d = 0; d < DH; ++d) {
		dk[d] = 0.f;
		dv[d] = 0.f;
	}
	for (int q0 = 0; q0 < a.Sq; q0 += TK) {
		__syncthreads();
		stage_rows<T, DH>(Qs, qb, a.ldq, q0, a.Sq, dh);
		stage_rows<T, DH>(Gs, gb, a.ldo, q0, a.Sq, dh);
		if (lane < TK) {
			const int qi = q0 + lane;
			float delta = 0.f, lse = 0.f;
			if (qi < a.Sq) {
				for (int d = 0; d < dh; ++d) delta += to_f32<T>(gb[(int64_t)qi * a.ldo + d]) * to_f32<T>(ob[(int64_t)qi * a.ldo + d]);
				lse = a.lse[((int64_t)b * a.n_head + h) * a.Sq + qi];
			}
			Ls[lane] = lse;
			Ds[lane] = delta;
			Rk[lane] = attn_row_key(attn_head_key(a, b * a.n_head + h), qi);
		}
		__syncthreads();
		const int nq = min(TK, a.Sq - q0);
		for (int j = 0; j < nq; ++j) {
			float s = 0.f, dp = 0.f;
#pragma unroll
			for (int d = 0; d < DH; ++d) {
				s += Qs[j][d] * Ks[lane][d];
				dp += Gs[j][d] * Vs[lane][d];
			}
			const float p = valid ? __expf(s * a.scale - Ls[j]) : 0.f;
			const float keep = drop ? attn_keep(attn_pair_bits(Rk[j], ki >> 1), ki, a.drop_thr16, a.drop_scale) : 1.f;
			const float ds = p * (dp * keep - Ds[j]) * a.scale;
			const float pv = p * keep;
#pragma unroll
			for (int d = 0; d < DH; ++d) {
				dv[d] += pv * Gs[j][d];
				dk[d] += ds * Qs[j][d];
			}
		}
	}
	if (valid) {
		T* dkp = reinterpret_cast<T*>(a.dk) + ((int64_t)b * a.Skv + ki) * a.lddk + h * dh;
		T* dvp = reinterpret_cast<T*>(a.dv) + ((int64_t)b * a.Skv + ki) * a.lddv + h * dh;
#pragma unroll
		for (int d = 0; d < DH; ++d)
			if (d < dh) {
				dkp[d] = from_f32<T>(dk[d]);
				dvp[d] = from_f32<T>(dv[d]);
			}
	}
}

template <typename T, int DH>
int launch_generic(const AttnArgs& a, bool bwd, hipStream_t st) {
	if (!bwd) {
		dim3 grid((unsigned)vitk_cdiv(a.Sq, 64), (unsigned)a.n_head, (unsigned)a.B);
		hipLaunchKernelGGL((attn_fwd_generic_kernel<T, DH>), grid, dim3(64), 0, st, a);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	}
	dim3 gq((unsigned)vitk_cdiv(a.Sq, 64), (unsigned)a.n_head, (unsigned)a.B);
	hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DH>), gq, dim3(64), 0, st, a);
	VITK_CHECK_LAUNCH();
	dim3 gk((unsigned)vitk_cdiv(a.Skv, 64), (unsigned)a.n_head, (unsigned)a.B);
	hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DH>), gk, dim3(64), 0, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

template <typename T>
int dispatch_dh(const AttnArgs& a, bool bwd, hipStream_t st) {
	if (a.dh <= 16) return launch_generic<T, 16>(a, bwd, st);
	if (a.dh <= 32) return launch_generic<T, 32>(a, bwd, st);
	if (a.dh <= 64) return launch_generic<T, 64>(a, bwd, st);
	return VITK_ERR_SHAPE;
}

}  // namespace

int vitk_attention_generic(const AttnArgs& a, bool bwd, int dtype, hipStream_t st) {
	if (dtype == VITK_F32) return dispatch_dh<float>(a, bwd, st);
	if (dtype == VITK_BF16) return dispatch_dh<bf16_t>(a, bwd, st);
	return VITK_ERR_DTYPE;
}
