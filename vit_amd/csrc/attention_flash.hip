// bf16 multi-head attention on the CDNA4 matrix cores for ANY sequence length and any head size <= 64 that is a multiple
// of 4 (F.scaled_dot_product_attention of layers/attention.py:39-43,84-88 in the reference): keys and values are walked in
// blocks of 64 with an online softmax, so the conv + attention hybrid's 784 tokens at head size 12 and 196 tokens at head
// size 24, and CvT's 3136-token first stage at 224x224, run on MFMA instead of the vector-ALU kernels.  (Head size 64 with
// at most 256 keys has its own kernels, which keep the whole key set in LDS.)
//
// Layout of every operand tile in LDS: 64 rows x 64 bf16 columns in 128-byte rows, 16-byte chunk index XORed with
// (row & 7) -- conflict-free for the row reads (ds_read_b128) and the transposing reads (ds_read_b64_tr_b16).  Columns past
// the head size are zero (the images are cleared once; staging only writes the head's columns), which pads the
// contraction of Q K^T to 32 or 64 and leaves the unused 16-column tiles of the P V product out.
//
// Orientation (as in the short-sequence kernels): mfma16(X, Y) gives element r of lane l = sum_k X[4 (l >> 4) + r][k] Y[l & 15][k].
//   forward / dQ (query on the lane):  S^T = K Q^T, softmax statistics per lane (+ 2 shuffles over the 4 lane groups),
//       O^T = V^T P^T with P^T taken from the accumulators of S^T as they stand (two 16-key tiles = one 32-deep k-step; the
//       transposing read of V uses the matching key order)
//   dK / dV (key on the lane):  S = Q K^T, dP = dO V^T, dV^T = dO^T P, dK^T = Q^T dS over all query blocks
// The backward is two kernels so that no gradient needs a cross-workgroup sum.  Dropout on the probabilities (DROP): the
// normaliser and lse are those of the undropped softmax; O, dP and the P that feeds dV carry the mask.
#include <cstdlib>

#include "attention.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int IMG = 64 * 128;  // one 64-row image

__device__ __forceinline__ int img_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ bf16x8 row_frag(const char* img, int row0, int s, int lane) {
	return *reinterpret_cast<const bf16x8*>(img + img_off(row0 + (lane & 15), 4 * s + (lane >> 4)));
}

// element j of lane (i = lane & 15, g = lane >> 4) = img[row(j)][col0 + i], row(j) = rowA + 4 g + j (j < 4), rowB + 4 g + j - 4
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int rowA, int rowB, int col0, int lane) {
	const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
	const int chunk = (col0 >> 3) + (p >> 1);
	const int sub = (p & 1) << 3;
	typedef s16x4 __attribute__((address_space(3))) * lds_ptr_t;
	const int ra = rowA + 4 * g + q, rb = rowB + 4 * g + q;
	s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(img + img_off(ra, chunk) + sub));
	s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(img + img_off(rb, chunk) + sub));
	union {
		s16x4 h[2];
		bf16x8 v;
	} u;
	u.h[0] = lo;
	u.h[1] = hi;
	return u.v;
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
	bf16x8 r;
	r[0] = (bf16_t)a[0]; r[1] = (bf16_t)a[1]; r[2] = (bf16_t)a[2]; r[3] = (bf16_t)a[3];
	r[4] = (bf16_t)b[0]; r[5] = (bf16_t)b[1]; r[6] = (bf16_t)b[2]; r[7] = (bf16_t)b[3];
	return r;
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ float group_max(float v) {  // over the 4 lane groups that share (lane & 15)
	v = fmaxf(v, __shfl_xor(v, 16, 64));
	return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
	v += __shfl_xor(v, 16, 64);
	return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ void zero_lds(char* p, int bytes) {
	for (int i = threadIdx.x * 16; i < bytes; i += blockDim.x * 16) *reinterpret_cast<uint4*>(p + i) = make_uint4(0, 0, 0, 0);
}

// Rows row0 .. row0 + 63 of a head's [total, dh] slice (row stride ld) into an image, 8 bytes at a time; rows past the end
// re-read the last row (finite values that only ever meet zero probabilities or unstored results).  dh4 = dh / 4.
__device__ __forceinline__ void stage_rows(char* img, const bf16_t* src, int64_t ld, int row0, int total, int dh4) {
	for (int idx = threadIdx.x; idx < 64 * dh4; idx += blockDim.x) {
		const int r = idx / dh4, c4 = idx - r * dh4;
		const int rs = min(row0 + r, total - 1);
		const uint2 v = *reinterpret_cast<const uint2*>(src + (int64_t)rs * ld + 4 * c4);
		*reinterpret_cast<uint2*>(img + img_off(r, c4 >> 1) + ((c4 & 1) << 3)) = v;
	}
}

// The same staging split in two so that a block's global loads fly while the previous block is computed on: the loads
// into registers (NIT = ceil(64 * dh4 / 256) 8-byte pieces per thread), and later the stores into the image.
template <int NIT>
__device__ __forceinline__ void fetch_rows(uint2 (&r)[NIT], const bf16_t* src, int64_t ld, int row0, int total, int dh4) {
#pragma unroll
	for (int i = 0; i < NIT; ++i) {
		const int idx = threadIdx.x + 256 * i;
		if (idx < 64 * dh4) {
			const int row = idx / dh4, c4 = idx - row * dh4;
			r[i] = *reinterpret_cast<const uint2*>(src + (int64_t)min(row0 + row, total - 1) * ld + 4 * c4);
		}
	}
}
template <int NIT>
__device__ __forceinline__ void commit_rows(char* img, const uint2 (&r)[NIT], int dh4) {
#pragma unroll
	for (int i = 0; i < NIT; ++i) {
		const int idx = threadIdx.x + 256 * i;
		if (idx < 64 * dh4) {
			const int row = idx / dh4, c4 = idx - row * dh4;
			*reinterpret_cast<uint2*>(img + img_off(row, c4 >> 1) + ((c4 & 1) << 3)) = r[i];
		}
	}
}

// 4 consecutive head columns d0 .. d0 + 3 of one row, if they belong to the head
__device__ __forceinline__ void store4(bf16_t* row, int d0, int dh, const f32x4& v, float mul) {
	if (d0 < dh) {
		bf16x4 o;
#pragma unroll
		for (int r = 0; r < 4; ++r) o[r] = (bf16_t)(v[r] * mul);
		*reinterpret_cast<bf16x4*>(row + d0) = o;
	}
}

// ---- forward: one workgroup = 64 queries of one (batch, head), one wave = 16 of them -----------------------------
// KS: 32-deep k-steps of Q K^T (head size padded to 32 KS); NDT: 16-column tiles of the output; DROP: dropout on the
// probabilities, the mask regenerated from a hash of (seed, batch * n_head + head, query, key pair) as in the other families
template <int KS, int NDT, bool DROP>
__global__ __launch_bounds__(256) void attn_flash_fwd_kernel(AttnArgs a) {
	if constexpr (DROP) attn_enter(a);
	__shared__ __attribute__((aligned(16))) char smem[3 * IMG];
	char *Qs = smem, *Ks = smem + IMG, *Vs = smem + 2 * IMG;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
	const int bh = blockIdx.y, b = bh / a.n_head, h = bh - b * a.n_head;
	const int q0 = blockIdx.x * 64;
	const int dh4 = a.dh >> 2;
	const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + h * a.dh;
	const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Skv * a.ldk + h * a.dh;
	const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Skv * a.ldv + h * a.dh;
	zero_lds(smem, 3 * IMG);
	__syncthreads();
	stage_rows(Qs, qb, a.ldq, q0, a.Sq, dh4);
	__syncthreads();
	bf16x8 qf[KS];
#pragma unroll
	for (int s = 0; s < KS; ++s) qf[s] = row_frag(Qs, 16 * wave, s, lane);
	const float c = a.scale * LOG2E;
	const uint32_t rk = DROP ? attn_row_key(attn_head_key(a, bh), q0 + 16 * wave + (lane & 15)) : 0u;
	float m = -INFINITY, l = 0.f;  // running maximum (log2 units of the scaled scores) and normaliser of this lane's query
	f32x4 o[NDT];
#pragma unroll
	for (int dt = 0; dt < NDT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
	uint2 kr[NDT], vr[NDT];
	fetch_rows(kr, kb, a.ldk, 0, a.Skv, dh4);
	fetch_rows(vr, vb, a.ldv, 0, a.Skv, dh4);
	for (int k0 = 0; k0 < a.Skv; k0 += 64) {
		__syncthreads();  // every wave is done with the previous block's images
		commit_rows(Ks, kr, dh4);
		commit_rows(Vs, vr, dh4);
		__syncthreads();
		if (k0 + 64 < a.Skv) {  // the next block's rows are on their way while this one is worked on
			fetch_rows(kr, kb, a.ldk, k0 + 64, a.Skv, dh4);
			fetch_rows(vr, vb, a.ldv, k0 + 64, a.Skv, dh4);
		}
		f32x4 st[4];
		float bm = -INFINITY;
#pragma unroll
		for (int t = 0; t < 4; ++t) {
			f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int s = 0; s < KS; ++s) acc = mfma16(row_frag(Ks, 16 * t, s, lane), qf[s], acc);
			if (k0 + 16 * t + 16 > a.Skv) {  // wave-uniform
#pragma unroll
				for (int r = 0; r < 4; ++r)
					if (k0 + 16 * t + 4 * g + r >= a.Skv) acc[r] = -INFINITY;
			}
#pragma unroll
			for (int r = 0; r < 4; ++r) bm = fmaxf(bm, acc[r]);
			st[t] = acc;
		}
		bm = group_max(bm);  // finite: every block holds at least one real key
		const float mn = fmaxf(m, bm * c);
		const float alpha = __builtin_amdgcn_exp2f(m - mn);  // 0 for the first block (m = -inf)
		float ls = 0.f;
#pragma unroll
		for (int t = 0; t < 4; ++t)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const float p = __builtin_amdgcn_exp2f(st[t][r] * c - mn);
				st[t][r] = p;
				ls += p;
			}
		l = l * alpha + group_sum(ls);  // (the normaliser is that of the undropped probabilities)
		m = mn;
		if constexpr (DROP) {
#pragma unroll
			for (int t = 0; t < 4; ++t) {
				const int kp = (k0 + 16 * t + 4 * g) >> 1;
				const uint32_t b0 = attn_pair_bits(rk, kp), b1 = attn_pair_bits(rk, kp + 1);
				st[t][0] *= attn_keep(b0, 0, a.drop_thr16, 1.f);
				st[t][1] *= attn_keep(b0, 1, a.drop_thr16, 1.f);
				st[t][2] *= attn_keep(b1, 0, a.drop_thr16, 1.f);
				st[t][3] *= attn_keep(b1, 1, a.drop_thr16, 1.f);
			}
		}
#pragma unroll
		for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
#pragma unroll
		for (int u = 0; u < 2; ++u) {
			const bf16x8 pf = pack8(st[2 * u], st[2 * u + 1]);
#pragma unroll
			for (int dt = 0; dt < NDT; ++dt) o[dt] = mfma16(tr_frag(Vs, 32 * u, 32 * u + 16, 16 * dt, lane), pf, o[dt]);
		}
	}
	const int q = q0 + 16 * wave + (lane & 15);
	if (q < a.Sq) {
		const float inv = (DROP ? a.drop_scale : 1.f) / l;
		bf16_t* op = reinterpret_cast<bf16_t*>(a.out_w) + ((int64_t)b * a.Sq + q) * a.ldo + h * a.dh;
#pragma unroll
		for (int dt = 0; dt < NDT; ++dt) store4(op, 16 * dt + 4 * g, a.dh, o[dt], inv);
		if (g == 0) a.lse[(int64_t)bh * a.Sq + q] = (m + __log2f(l)) * (1.f / LOG2E);
	}
}

// ---- backward, dQ: one wave = 16 queries; also leaves lse (log2 units) and delta = rowsum(dO o O) in the scratch --------
template <int KS, int NDT, bool DROP>
__global__ __launch_bounds__(256) void attn_flash_dq_kernel(AttnArgs a) {
	if constexpr (DROP) attn_enter(a);
	__shared__ __attribute__((aligned(16))) char smem[4 * IMG];
	char *Qs = smem, *Gs = smem + IMG, *Ks = smem + 2 * IMG, *Vs = smem + 3 * IMG;  // Ks doubles as the O image first
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
	const int bh = blockIdx.y, b = bh / a.n_head, h = bh - b * a.n_head;
	const int n_bh = a.B * a.n_head;
	const int q0 = blockIdx.x * 64;
	const int dh4 = a.dh >> 2;
	const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Skv * a.ldk + h * a.dh;
	const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Skv * a.ldv + h * a.dh;
	zero_lds(smem, 4 * IMG);
	__syncthreads();
	stage_rows(Qs, reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + h * a.dh, a.ldq, q0, a.Sq, dh4);
	stage_rows(Gs, reinterpret_cast<const bf16_t*>(a.dout) + (int64_t)b * a.Sq * a.ldo + h * a.dh, a.ldo, q0, a.Sq, dh4);
	stage_rows(Ks, reinterpret_cast<const bf16_t*>(a.out) + (int64_t)b * a.Sq * a.ldo + h * a.dh, a.ldo, q0, a.Sq, dh4);
	__syncthreads();
	bf16x8 qf[KS], gf[KS];
	float delta = 0.f;
#pragma unroll
	for (int s = 0; s < KS; ++s) {
		qf[s] = row_frag(Qs, 16 * wave, s, lane);
		gf[s] = row_frag(Gs, 16 * wave, s, lane);
		const bf16x8 of = row_frag(Ks, 16 * wave, s, lane);
#pragma unroll
		for (int j = 0; j < 8; ++j) delta += (float)gf[s][j] * (float)of[j];
	}
	delta = group_sum(delta);
	const int q = q0 + 16 * wave + (lane & 15);
	const int qc = min(q, a.Sq - 1);
	const float lse2 = a.lse[(int64_t)bh * a.Sq + qc] * LOG2E;
	if (g == 0 && q < a.Sq) {
		a.bwd_scratch[(int64_t)bh * a.Sq + q] = lse2;
		a.bwd_scratch[((int64_t)n_bh + bh) * a.Sq + q] = delta;
	}
	const float c = a.scale * LOG2E;
	const uint32_t rk = DROP ? attn_row_key(attn_head_key(a, bh), q) : 0u;
	f32x4 dq[NDT];
#pragma unroll
	for (int dt = 0; dt < NDT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
	uint2 kr[NDT], vr[NDT];
	fetch_rows(kr, kb, a.ldk, 0, a.Skv, dh4);
	fetch_rows(vr, vb, a.ldv, 0, a.Skv, dh4);
	for (int k0 = 0; k0 < a.Skv; k0 += 64) {
		__syncthreads();  // (first pass: every wave has taken its O fragments out of the image that now receives K)
		commit_rows(Ks, kr, dh4);
		commit_rows(Vs, vr, dh4);
		__syncthreads();
		if (k0 + 64 < a.Skv) {
			fetch_rows(kr, kb, a.ldk, k0 + 64, a.Skv, dh4);
			fetch_rows(vr, vb, a.ldv, k0 + 64, a.Skv, dh4);
		}
#pragma unroll
		for (int u = 0; u < 2; ++u) {
			f32x4 ds[2];
#pragma unroll
			for (int hf = 0; hf < 2; ++hf) {
				const int t = 2 * u + hf;
				f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int ks = 0; ks < KS; ++ks) {
					s = mfma16(row_frag(Ks, 16 * t, ks, lane), qf[ks], s);
					dp = mfma16(row_frag(Vs, 16 * t, ks, lane), gf[ks], dp);
				}
				if constexpr (DROP) {
					const int kp = (k0 + 16 * t + 4 * g) >> 1;
					const uint32_t b0 = attn_pair_bits(rk, kp), b1 = attn_pair_bits(rk, kp + 1);
					dp[0] *= attn_keep(b0, 0, a.drop_thr16, a.drop_scale);
					dp[1] *= attn_keep(b0, 1, a.drop_thr16, a.drop_scale);
					dp[2] *= attn_keep(b1, 0, a.drop_thr16, a.drop_scale);
					dp[3] *= attn_keep(b1, 1, a.drop_thr16, a.drop_scale);
				}
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const float p = __builtin_amdgcn_exp2f(s[r] * c - lse2);
					ds[hf][r] = (k0 + 16 * t + 4 * g + r < a.Skv) ? p * (dp[r] - delta) * a.scale : 0.f;
				}
			}
			const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
			for (int dt = 0; dt < NDT; ++dt) dq[dt] = mfma16(tr_frag(Ks, 32 * u, 32 * u + 16, 16 * dt, lane), dsf, dq[dt]);
		}
	}
	if (q < a.Sq) {
		bf16_t* dqp = reinterpret_cast<bf16_t*>(a.dq) + ((int64_t)b * a.Sq + q) * a.lddq + h * a.dh;
#pragma unroll
		for (int dt = 0; dt < NDT; ++dt) store4(dqp, 16 * dt + 4 * g, a.dh, dq[dt], 1.f);
	}
}

// ---- backward, dK / dV: one wave = 16 keys, summed over all query blocks ------------------------------------------------
template <int KS, int NDT, bool DROP>
__global__ __launch_bounds__(256) void attn_flash_dkv_kernel(AttnArgs a) {
	if constexpr (DROP) attn_enter(a);
	__shared__ __attribute__((aligned(16))) char smem[4 * IMG + 3 * 64 * 4];
	char *Ks = smem, *Vs = smem + IMG, *Qs = smem + 2 * IMG, *Gs = smem + 3 * IMG;
	float* lse_s = reinterpret_cast<float*>(smem + 4 * IMG);
	float* delta_s = lse_s + 64;
	uint32_t* rk_s = reinterpret_cast<uint32_t*>(delta_s + 64);  // DROP: mask row keys of the block's 64 queries
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
	const int bh = blockIdx.y, b = bh / a.n_head, h = bh - b * a.n_head;
	const int n_bh = a.B * a.n_head;
	const int k0 = blockIdx.x * 64;
	const int dh4 = a.dh >> 2;
	const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + h * a.dh;
	const bf16_t* gb = reinterpret_cast<const bf16_t*>(a.dout) + (int64_t)b * a.Sq * a.ldo + h * a.dh;
	zero_lds(smem, 4 * IMG);
	__syncthreads();
	stage_rows(Ks, reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Skv * a.ldk + h * a.dh, a.ldk, k0, a.Skv, dh4);
	stage_rows(Vs, reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Skv * a.ldv + h * a.dh, a.ldv, k0, a.Skv, dh4);
	__syncthreads();
	bf16x8 kf[KS], vf[KS];
#pragma unroll
	for (int s = 0; s < KS; ++s) {
		kf[s] = row_frag(Ks, 16 * wave, s, lane);
		vf[s] = row_frag(Vs, 16 * wave, s, lane);
	}
	const float c = a.scale * LOG2E;
	f32x4 dk[NDT], dv[NDT];
#pragma unroll
	for (int dt = 0; dt < NDT; ++dt) dk[dt] = dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
	uint2 qr[NDT], gr[NDT];
	float stat = 0.f;  // threads 0..63: lse of query q0 + t; 64..127: delta of query q0 + t - 64
	auto fetch_stat = [&](int q0) {
		if (threadIdx.x < 128) {
			const int q = q0 + (threadIdx.x & 63);
			const bool is_lse = threadIdx.x < 64;
			// probabilities of padded queries: exp2(s - inf) = 0
			stat = q < a.Sq ? a.bwd_scratch[((int64_t)(is_lse ? 0 : n_bh) + bh) * a.Sq + q] : (is_lse ? INFINITY : 0.f);
		}
	};
	fetch_rows(qr, qb, a.ldq, 0, a.Sq, dh4);
	fetch_rows(gr, gb, a.ldo, 0, a.Sq, dh4);
	fetch_stat(0);
	for (int q0 = 0; q0 < a.Sq; q0 += 64) {
		__syncthreads();
		commit_rows(Qs, qr, dh4);
		commit_rows(Gs, gr, dh4);
		if (threadIdx.x < 128) lse_s[threadIdx.x] = stat;  // delta_s follows lse_s
		if constexpr (DROP) {
			if (threadIdx.x >= 192) rk_s[threadIdx.x - 192] = attn_row_key(attn_head_key(a, bh), q0 + (int)threadIdx.x - 192);
		}
		__syncthreads();
		if (q0 + 64 < a.Sq) {
			fetch_rows(qr, qb, a.ldq, q0 + 64, a.Sq, dh4);
			fetch_rows(gr, gb, a.ldo, q0 + 64, a.Sq, dh4);
			fetch_stat(q0 + 64);
		}
#pragma unroll
		for (int u = 0; u < 2; ++u) {
			f32x4 pr[2], ds[2];
#pragma unroll
			for (int hf = 0; hf < 2; ++hf) {
				const int t = 2 * u + hf;
				f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int ks = 0; ks < KS; ++ks) {
					s = mfma16(row_frag(Qs, 16 * t, ks, lane), kf[ks], s);
					dp = mfma16(row_frag(Gs, 16 * t, ks, lane), vf[ks], dp);
				}
				const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * t + 4 * g);
				const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta_s + 16 * t + 4 * g);
				float keep[4] = {1.f, 1.f, 1.f, 1.f};
				if constexpr (DROP) {
					const int key = k0 + 16 * wave + (lane & 15);
					const uint4 rk4 = *reinterpret_cast<const uint4*>(rk_s + 16 * t + 4 * g);
					keep[0] = attn_keep(attn_pair_bits(rk4.x, key >> 1), key, a.drop_thr16, a.drop_scale);
					keep[1] = attn_keep(attn_pair_bits(rk4.y, key >> 1), key, a.drop_thr16, a.drop_scale);
					keep[2] = attn_keep(attn_pair_bits(rk4.z, key >> 1), key, a.drop_thr16, a.drop_scale);
					keep[3] = attn_keep(attn_pair_bits(rk4.w, key >> 1), key, a.drop_thr16, a.drop_scale);
				}
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const float p = __builtin_amdgcn_exp2f(s[r] * c - l4[r]);
					pr[hf][r] = DROP ? p * keep[r] : p;
					ds[hf][r] = p * ((DROP ? dp[r] * keep[r] : dp[r]) - d4[r]) * a.scale;
				}
			}
			const bf16x8 pf = pack8(pr[0], pr[1]), dsf = pack8(ds[0], ds[1]);
#pragma unroll
			for (int dt = 0; dt < NDT; ++dt) {
				dv[dt] = mfma16(tr_frag(Gs, 32 * u, 32 * u + 16, 16 * dt, lane), pf, dv[dt]);
				dk[dt] = mfma16(tr_frag(Qs, 32 * u, 32 * u + 16, 16 * dt, lane), dsf, dk[dt]);
			}
		}
	}
	const int key = k0 + 16 * wave + (lane & 15);
	if (key < a.Skv) {
		bf16_t* dkp = reinterpret_cast<bf16_t*>(a.dk) + ((int64_t)b * a.Skv + key) * a.lddk + h * a.dh;
		bf16_t* dvp = reinterpret_cast<bf16_t*>(a.dv) + ((int64_t)b * a.Skv + key) * a.lddv + h * a.dh;
#pragma unroll
		for (int dt = 0; dt < NDT; ++dt) {
			store4(dkp, 16 * dt + 4 * g, a.dh, dk[dt], 1.f);
			store4(dvp, 16 * dt + 4 * g, a.dh, dv[dt], 1.f);
		}
	}
}

// ---- backward in ONE pass (small heads): lse (log2 units) and delta = rowsum(dO o O) first ... --------------------------------
// one thread per (batch, query, head), head fastest: neighbouring threads read neighbouring head slices of one token row
__global__ __launch_bounds__(256) void attn_flash_delta_kernel(AttnArgs a) {
	const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
	const int64_t n_bh = (int64_t)a.B * a.n_head;
	if (idx >= n_bh * a.Sq) return;
	const int h = (int)(idx % a.n_head);
	const int64_t bq = idx / a.n_head;
	const int q = (int)(bq % a.Sq), b = (int)(bq / a.Sq);
	const bf16_t* op = reinterpret_cast<const bf16_t*>(a.out) + ((int64_t)b * a.Sq + q) * a.ldo + h * a.dh;
	const bf16_t* gp = reinterpret_cast<const bf16_t*>(a.dout) + ((int64_t)b * a.Sq + q) * a.ldo + h * a.dh;
	float delta = 0.f;
	for (int d = 0; d < a.dh; d += 4) {
		const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(op + d), g4 = *reinterpret_cast<const bf16x4*>(gp + d);
#pragma unroll
		for (int j = 0; j < 4; ++j) delta += (float)o4[j] * (float)g4[j];
	}
	const int64_t bh = (int64_t)b * a.n_head + h;
	a.bwd_scratch[bh * a.Sq + q] = a.lse[bh * a.Sq + q] * LOG2E;
	a.bwd_scratch[(n_bh + bh) * a.Sq + q] = delta;
}

// ... then one workgroup per (batch, head): keys in blocks of 64 (outer loop: a wave owns 16 keys, their K / V fragments and
// dK / dV accumulators in registers, key on the lane as in the dK / dV kernel), queries in blocks of 64 (inner loop).  S, dP, P and
// dS of a (key block, query block) pair are formed ONCE -- the two-kernel form computes them twice, and at head size 12 that
// softmax arithmetic, not the matrix pipe, is what the time goes into.  dS goes through an LDS image [key][query] (the V image's
// space: V lives in registers by then) and comes back by transposing reads as the operand of dQ^T += K^T dS^T, query on the
// lane; dQ of all the head's queries accumulates in a float32 LDS array [Sq][dh] (a wave only ever touches the rows of its own
// 16 queries of a block: plain read-add-write), and leaves once at the end.  No gradient needs a cross-workgroup sum.
template <int KS, int NDT, bool DROP>
__global__ __launch_bounds__(256) void attn_flash_bwd1_kernel(AttnArgs a) {
	if constexpr (DROP) attn_enter(a);
	extern __shared__ __attribute__((aligned(16))) char smem[];
	char *Ks = smem, *Vs = smem + IMG, *Qs = smem + 2 * IMG, *Gs = smem + 3 * IMG;
	char* DSs = Vs;
	float* lse_s = reinterpret_cast<float*>(smem + 4 * IMG);
	float* delta_s = lse_s + 64;
	uint32_t* rk_s = reinterpret_cast<uint32_t*>(delta_s + 64);
	float* dq_s = reinterpret_cast<float*>(smem + 4 * IMG + 3 * 64 * 4);  // [Sq][dh]
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
	const int bh = blockIdx.x, b = bh / a.n_head, h = bh - b * a.n_head;
	const int n_bh = a.B * a.n_head;
	const int dh4 = a.dh >> 2;
	const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + h * a.dh;
	const bf16_t* gb = reinterpret_cast<const bf16_t*>(a.dout) + (int64_t)b * a.Sq * a.ldo + h * a.dh;
	const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Skv * a.ldk + h * a.dh;
	const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Skv * a.ldv + h * a.dh;
	zero_lds(smem, 4 * IMG);
	for (int i = threadIdx.x; i < a.Sq * a.dh; i += 256) dq_s[i] = 0.f;
	const float c = a.scale * LOG2E;
	for (int k0 = 0; k0 < a.Skv; k0 += 64) {
		__syncthreads();  // the previous key block's readers are done with every image (and the fills above are complete)
		if (k0 > 0) zero_lds(Vs, IMG);  // (it held dS values in columns past the head size)
		__syncthreads();
		stage_rows(Ks, kb, a.ldk, k0, a.Skv, dh4);
		stage_rows(Vs, vb, a.ldv, k0, a.Skv, dh4);
		__syncthreads();
		bf16x8 kf[KS], vf[KS];
#pragma unroll
		for (int s = 0; s < KS; ++s) {
			kf[s] = row_frag(Ks, 16 * wave, s, lane);
			vf[s] = row_frag(Vs, 16 * wave, s, lane);
		}
		const int key = k0 + 16 * wave + (lane & 15);
		const bool key_ok = key < a.Skv;
		f32x4 dk[NDT], dv[NDT];
#pragma unroll
		for (int dt = 0; dt < NDT; ++dt) dk[dt] = dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
		uint2 qr[NDT], gr[NDT];
		float stat = 0.f;  // threads 0..63: lse of query q0 + t; 64..127: delta of query q0 + t - 64
		auto fetch_stat = [&](int q0) {
			if (threadIdx.x < 128) {
				const int q = q0 + (threadIdx.x & 63);
				const bool is_lse = threadIdx.x < 64;
				stat = q < a.Sq ? a.bwd_scratch[((int64_t)(is_lse ? 0 : n_bh) + bh) * a.Sq + q] : (is_lse ? INFINITY : 0.f);  // padded queries: probability 0
			}
		};
		fetch_rows(qr, qb, a.ldq, 0, a.Sq, dh4);
		fetch_rows(gr, gb, a.ldo, 0, a.Sq, dh4);
		fetch_stat(0);
		for (int q0 = 0; q0 < a.Sq; q0 += 64) {
			__syncthreads();  // every wave has its V fragments / is done with the previous query block's Q, dO and dS images
			commit_rows(Qs, qr, dh4);
			commit_rows(Gs, gr, dh4);
			if (threadIdx.x < 128) lse_s[threadIdx.x] = stat;  // delta_s follows lse_s
			if constexpr (DROP) {
				if (threadIdx.x >= 192) rk_s[threadIdx.x - 192] = attn_row_key(attn_head_key(a, bh), q0 + (int)threadIdx.x - 192);
			}
			__syncthreads();
			if (q0 + 64 < a.Sq) {
				fetch_rows(qr, qb, a.ldq, q0 + 64, a.Sq, dh4);
				fetch_rows(gr, gb, a.ldo, q0 + 64, a.Sq, dh4);
				fetch_stat(q0 + 64);
			}
#pragma unroll
			for (int u = 0; u < 2; ++u) {
				f32x4 pr[2], ds[2];
#pragma unroll
				for (int hf = 0; hf < 2; ++hf) {
					const int t = 2 * u + hf;
					f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
					for (int ks = 0; ks < KS; ++ks) {
						s = mfma16(row_frag(Qs, 16 * t, ks, lane), kf[ks], s);
						dp = mfma16(row_frag(Gs, 16 * t, ks, lane), vf[ks], dp);
					}
					const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * t + 4 * g);
					const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta_s + 16 * t + 4 * g);
					float keep[4] = {1.f, 1.f, 1.f, 1.f};
					if constexpr (DROP) {
						const uint4 rk4 = *reinterpret_cast<const uint4*>(rk_s + 16 * t + 4 * g);
						keep[0] = attn_keep(attn_pair_bits(rk4.x, key >> 1), key, a.drop_thr16, a.drop_scale);
						keep[1] = attn_keep(attn_pair_bits(rk4.y, key >> 1), key, a.drop_thr16, a.drop_scale);
						keep[2] = attn_keep(attn_pair_bits(rk4.z, key >> 1), key, a.drop_thr16, a.drop_scale);
						keep[3] = attn_keep(attn_pair_bits(rk4.w, key >> 1), key, a.drop_thr16, a.drop_scale);
					}
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						const float p = key_ok ? __builtin_amdgcn_exp2f(s[r] * c - l4[r]) : 0.f;  // (a key past the last must not reach dQ)
						pr[hf][r] = DROP ? p * keep[r] : p;
						ds[hf][r] = p * ((DROP ? dp[r] * keep[r] : dp[r]) - d4[r]) * a.scale;
					}
					// dS of this lane's key, queries 16 t + 4 g .. + 3, into the [key][query] image
					bf16x4 d16;
#pragma unroll
					for (int r = 0; r < 4; ++r) d16[r] = (bf16_t)ds[hf][r];
					const int qc = 16 * t + 4 * g;
					*reinterpret_cast<bf16x4*>(DSs + img_off(16 * wave + (lane & 15), qc >> 3) + (((qc >> 2) & 1) << 3)) = d16;
				}
				const bf16x8 pf = pack8(pr[0], pr[1]), dsf = pack8(ds[0], ds[1]);
#pragma unroll
				for (int dt = 0; dt < NDT; ++dt) {
					dv[dt] = mfma16(tr_frag(Gs, 32 * u, 32 * u + 16, 16 * dt, lane), pf, dv[dt]);
					dk[dt] = mfma16(tr_frag(Qs, 32 * u, 32 * u + 16, 16 * dt, lane), dsf, dk[dt]);
				}
			}
			__syncthreads();  // the dS image is complete
			// dQ^T (this key block's part) of the wave's 16 queries: rows d, query on the lane
			const int q = q0 + 16 * wave + (lane & 15);
#pragma unroll
			for (int dt = 0; dt < NDT; ++dt) {
				f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
				for (int u = 0; u < 2; ++u) acc = mfma16(tr_frag(Ks, 32 * u, 32 * u + 16, 16 * dt, lane), tr_frag(DSs, 32 * u, 32 * u + 16, 16 * wave, lane), acc);
				const int d0 = 16 * dt + 4 * g;
				if (q < a.Sq && d0 < a.dh) {
					f32x4* dst = reinterpret_cast<f32x4*>(dq_s + q * a.dh + d0);
					*dst = *dst + acc;
				}
			}
		}
		if (key_ok) {
			bf16_t* dkp = reinterpret_cast<bf16_t*>(a.dk) + ((int64_t)b * a.Skv + key) * a.lddk + h * a.dh;
			bf16_t* dvp = reinterpret_cast<bf16_t*>(a.dv) + ((int64_t)b * a.Skv + key) * a.lddv + h * a.dh;
#pragma unroll
			for (int dt = 0; dt < NDT; ++dt) {
				store4(dkp, 16 * dt + 4 * g, a.dh, dk[dt], 1.f);
				store4(dvp, 16 * dt + 4 * g, a.dh, dv[dt], 1.f);
			}
		}
	}
	__syncthreads();
	bf16_t* dqb = reinterpret_cast<bf16_t*>(a.dq) + (int64_t)b * a.Sq * a.lddq + h * a.dh;
	for (int idx = threadIdx.x; idx < a.Sq * dh4; idx += 256) {
		const int q = idx / dh4, c4 = idx - q * dh4;
		const f32x4 v = *reinterpret_cast<const f32x4*>(dq_s + q * a.dh + 4 * c4);
		bf16x4 o;
#pragma unroll
		for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
		*reinterpret_cast<bf16x4*>(dqb + (int64_t)q * a.lddq + 4 * c4) = o;
	}
}

// LDS of the one-pass backward: four images, the per-block statistics, dQ of the whole head in float32
inline size_t bwd1_lds(const AttnArgs& a) { return (size_t)4 * IMG + 3 * 64 * 4 + (size_t)a.Sq * a.dh * 4; }

template <int KS, int NDT, bool DROP>
int run_d(const AttnArgs& a, bool bwd, hipStream_t st) {
	const unsigned n_bh = (unsigned)(a.B * a.n_head);
	if (!bwd) {
		hipLaunchKernelGGL((attn_flash_fwd_kernel<KS, NDT, DROP>), dim3((unsigned)vitk_cdiv(a.Sq, 64), n_bh), dim3(256), 0, st, a);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	}
	// small heads over a sequence whose dQ fits LDS beside the images with two workgroups per CU: the one-pass form
	static const bool one_pass = getenv("VITK_ATTN_FLASH_ONE_PASS") == nullptr || getenv("VITK_ATTN_FLASH_ONE_PASS")[0] != '0';  // A-B switch
	// (measured per layer of the conv + attention hybrid at batch 128: 196 tokens x 16 heads of 24: 558 -> 418 us; 784 tokens x 32
	// heads of 12: 2568 -> 2614 us -- with 13 key blocks the three barriers per (key block, query block) pair at two workgroups
	// per CU cost what the second softmax saved, so long sequences keep the two-kernel form)
	if (one_pass && NDT <= 2 && a.Skv <= 256 && bwd1_lds(a) <= 80 * 1024) {
		const size_t lds = bwd1_lds(a);
		static bool attr_set = false;
		if (!attr_set) {
			hipError_t e = hipFuncSetAttribute((const void*)attn_flash_bwd1_kernel<KS, NDT, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
			if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
			attr_set = true;
		}
		hipLaunchKernelGGL(attn_flash_delta_kernel, dim3((unsigned)vitk_cdiv((int64_t)n_bh * a.Sq, 256)), dim3(256), 0, st, a);
		VITK_CHECK_LAUNCH();
		hipLaunchKernelGGL((attn_flash_bwd1_kernel<KS, NDT, DROP>), dim3(n_bh), dim3(256), lds, st, a);
		VITK_CHECK_LAUNCH();
		return VITK_OK;
	}
	hipLaunchKernelGGL((attn_flash_dq_kernel<KS, NDT, DROP>), dim3((unsigned)vitk_cdiv(a.Sq, 64), n_bh), dim3(256), 0, st, a);
	VITK_CHECK_LAUNCH();
	hipLaunchKernelGGL((attn_flash_dkv_kernel<KS, NDT, DROP>), dim3((unsigned)vitk_cdiv(a.Skv, 64), n_bh), dim3(256), 0, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

template <int KS, int NDT>
int run(const AttnArgs& a, bool bwd, hipStream_t st) {
	return a.drop_thr16 != 0 ? run_d<KS, NDT, true>(a, bwd, st) : run_d<KS, NDT, false>(a, bwd, st);
}

bool al8(const void* p) { return p == nullptr || vitk_aligned(p, 8); }

}  // namespace

bool vitk_attention_flash_ok(const AttnArgs& a, int dtype) {
	if (dtype != VITK_BF16 || a.dh < 4 || a.dh > 64 || a.dh % 4 != 0) return false;
	if ((a.ldq | a.ldk | a.ldv | a.ldo) % 4 != 0) return false;
	if (a.dq != nullptr && ((a.lddq | a.lddk | a.lddv) % 4 != 0)) return false;
	if ((int64_t)a.B * a.n_head > 65535) return false;  // grid.y
	return al8(a.q) && al8(a.k) && al8(a.v) && al8(a.out) && al8(a.dout) && al8(a.out_w) && al8(a.dq) && al8(a.dk) && al8(a.dv);
}

int vitk_attention_flash(const AttnArgs& a, bool bwd, hipStream_t st) {
	if (bwd && a.bwd_scratch == nullptr) return VITK_ERR_NULL;
	const int ndt = (a.dh + 15) / 16;
	switch (ndt) {
		case 1: return run<1, 1>(a, bwd, st);
		case 2: return run<1, 2>(a, bwd, st);
		case 3: return run<2, 3>(a, bwd, st);
		default: return run<2, 4>(a, bwd, st);
	}
}
