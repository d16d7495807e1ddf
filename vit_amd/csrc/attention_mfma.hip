// bf16 multi-head attention on the CDNA4 matrix cores for short sequences (head size 64, up to 256 keys):
// the whole key/value set of one (batch, head) sits in LDS and the softmax of a query row is taken over
// all keys at once in registers (no online rescaling is needed at ViT sequence lengths).
//
// One workgroup = one (batch, head); one wavefront = one 16-row tile of queries (forward, dQ) or keys (dK/dV).
//
// Forward, per wave (q on the MFMA lane):
//   S^T[key][q] = K Q^T          A = K rows from LDS (ds_read_b128), B = Q rows straight from HBM
//   P^T = softmax over keys      keys live in the accumulator registers + 4 lane groups: 2 shuffles per reduction
//   O^T[d][q]  = V^T P^T         A = V^T by the transposing LDS read (ds_read_b64_tr_b16), B = P^T taken from the
//                                accumulators of the first product with no lane movement (two 16-key tiles = one
//                                32-deep k-step; the k -> key order of the A operand is permuted to match)
// Backward is two kernels with the same structure, so that no gradient needs a cross-workgroup sum:
//   dQ kernel (q on the lane):   S^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta) * scale, dQ^T = K^T dS^T
//   dK/dV kernel (key on lane):  S = Q K^T, dP = dO V^T, dV^T = dO^T P, dK^T = Q^T dS; wave owns 16 keys, sums over all q
//
// LDS images are [row][64] bf16 with 128-byte rows and the 16-byte chunk index XOR-ed with (row & 7): conflict-free
// both for the row reads (ds_read_b128) and for the transposing reads.
#include "attention.h"

namespace {

constexpr int DH = 64;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int img_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// A/B fragment of 16 rows starting at row0, k-step s (32 of the 64 head columns): lane -> row (lane & 15), chunk 4 s + (lane >> 4)
__device__ __forceinline__ bf16x8 row_frag(const char* img, int row0, int s, int lane) {
	return *reinterpret_cast<const bf16x8*>(img + img_off(row0 + (lane & 15), 4 * s + (lane >> 4)));
}

// Transposed fragment: element j of lane (i = lane & 15, g = lane >> 4) = img[row(j)][col0 + i] with
// row(j) = rowA + 4 g + j for j < 4 and rowB + 4 g + (j - 4) for j >= 4.
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

// Stage rows [0, rows_pad) x 64 columns of a [*, ld] bf16 matrix into an LDS image; rows >= rows_valid are zero.
__device__ __forceinline__ void stage_image(char* img, const bf16_t* src, int64_t ld, int rows_valid, int rows_pad) {
	for (int idx = threadIdx.x; idx < rows_pad * 8; idx += blockDim.x) {
		const int r = idx >> 3, c = idx & 7;
		uint4 v = make_uint4(0, 0, 0, 0);
		if (r < rows_valid) v = *reinterpret_cast<const uint4*>(src + (int64_t)r * ld + 8 * c);
		*reinterpret_cast<uint4*>(img + img_off(r, c)) = v;
	}
}

__device__ __forceinline__ float group_max(float v) {  // over the 4 lane groups that share (lane & 15)
	v = fmaxf(v, __shfl_xor(v, 16, 64));
	return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
	v += __shfl_xor(v, 16, 64);
	return v + __shfl_xor(v, 32, 64);
}

// ---------------------------------------------------------------------------------------------------------------
// staging by LDS-DMA
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

// Rows [0, rows_pad) x 64 columns of a [*, ld] bf16 matrix into an LDS image by global_load_lds_dwordx4: a
// wave-instruction fills 8 image rows (1 KiB, written linearly at a wave-uniform address); the image's chunk swizzle is
// applied to each lane's SOURCE address.  Rows past rows_valid re-read the last valid row (finite values that only ever
// meet zero probabilities).  Pieces are dealt round-robin to the waves; nothing is waited for here.
__device__ __forceinline__ void stage_image_dma(char* img, const bf16_t* src, int64_t ld, int rows_valid, int rows_pad, int wave, int nwaves, int lane) {
	const int r8 = lane >> 3, slot = lane & 7;
	for (int piece = wave; piece < (rows_pad >> 3); piece += nwaves) {
		const int row = 8 * piece + r8;
		const int rs = row < rows_valid ? row : rows_valid - 1;
		const bf16_t* p = src + (int64_t)rs * ld + 8 * (slot ^ (row & 7));
		__builtin_amdgcn_global_load_lds((gbl_void_p)p, (lds_void_p)(img + piece * 1024), 16, 0, 0);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
// A workgroup walks `a.heads_per_wg` consecutive (batch, head) pairs with two K/V image buffers: while it works on one
// pair the LDS-DMA of the next pair's keys and values is in flight, and each wave already holds the next pair's query
// rows in registers.  (One workgroup per pair spends most of its life waiting for its own K/V and Q to arrive: with
// everything else left out, the staging and launch skeleton alone took 100 of that kernel's 130 us on ViT-B/16.)
template <int KT2, bool DROP>  // keys padded to 32 * KT2; DROP: attention-probability dropout
__global__ __launch_bounds__(1024) void attn_fwd_mfma_kernel(AttnArgs a) {
	if constexpr (DROP) attn_enter(a);
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr int IMG = KT2 * 32 * 128;  // one image
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
	const int g = lane >> 4;
	const int n_bh = a.B * a.n_head;
	const int bh0 = blockIdx.x * a.heads_per_wg;
	const int bh1 = min(n_bh, bh0 + a.heads_per_wg);
	const float c = a.scale * LOG2E;
	const int n_qtiles = (a.Sq + 15) >> 4;  // <= number of waves: one query tile per wave and pair
	const bool active = wave < n_qtiles;
	const int q = wave * 16 + (lane & 15);
	const int qc = q < a.Sq ? q : a.Sq - 1;

	auto stage = [&](int bh, int buf) {
		const int b = bh / a.n_head, h = bh - b * a.n_head;
		const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Skv * a.ldk + h * DH;
		const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Skv * a.ldv + h * DH;
		stage_image_dma(smem + buf * 2 * IMG, kb, a.ldk, a.Skv, KT2 * 32, wave, nwaves, lane);
		stage_image_dma(smem + buf * 2 * IMG + IMG, vb, a.ldv, a.Skv, KT2 * 32, wave, nwaves, lane);
	};
	auto q_ptr = [&](int bh) {
		const int b = bh / a.n_head, h = bh - b * a.n_head;
		return reinterpret_cast<const bf16_t*>(a.q) + ((int64_t)b * a.Sq + qc) * a.ldq + h * DH + 8 * g;
	};

	bf16x8 qn0 = {}, qn1 = {};
	if (active) {
		const bf16_t* qp = q_ptr(bh0);
		qn0 = *reinterpret_cast<const bf16x8*>(qp);
		qn1 = *reinterpret_cast<const bf16x8*>(qp + 32);
	}
	stage(bh0, 0);
	for (int bh = bh0; bh < bh1; ++bh) {
		const int buf = (bh - bh0) & 1;
		// this pair's images have landed (the barrier's fence drains the DMA, and with it the query rows requested
		// before it) and every wave is done with the other buffer
		__syncthreads();
		const bf16x8 qf0 = qn0, qf1 = qn1;
		if (bh + 1 < bh1) {
			stage(bh + 1, buf ^ 1);
			if (active) {
				const bf16_t* qp = q_ptr(bh + 1);
				qn0 = *reinterpret_cast<const bf16x8*>(qp);
				qn1 = *reinterpret_cast<const bf16x8*>(qp + 32);
			}
		}
		if (!active) continue;
		const char* Ks = smem + buf * 2 * IMG;
		const char* Vs = Ks + IMG;
		const int b = bh / a.n_head, h = bh - b * a.n_head;
		f32x4 st[2 * KT2];
		float m = -INFINITY;
#pragma unroll
		for (int t = 0; t < 2 * KT2; ++t) {
			f32x4 acc = {0.f, 0.f, 0.f, 0.f};
			acc = mfma16(row_frag(Ks, 16 * t, 0, lane), qf0, acc);
			acc = mfma16(row_frag(Ks, 16 * t, 1, lane), qf1, acc);
			if (t >= 2 * KT2 - 2) {  // 32 (KT2 - 1) < Skv <= 32 KT2: only the last two tiles can reach past the last key
#pragma unroll
				for (int r = 0; r < 4; ++r)
					if (16 * t + 4 * g + r >= a.Skv) acc[r] = -INFINITY;
			}
#pragma unroll
			for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[r]);
			st[t] = acc;
			__builtin_amdgcn_sched_barrier(0);  // keep the unrolled iterations in order: hoisting all LDS reads spills
		}
		m = group_max(m);
		const float mc = m * c;
		float l = 0.f;
#pragma unroll
		for (int t = 0; t < 2 * KT2; ++t)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const float p = __builtin_amdgcn_exp2f(st[t][r] * c - mc);
				st[t][r] = p;
				l += p;
			}
		l = group_sum(l);
		// dropout: the normaliser l is that of the full softmax; dropped probabilities leave the P V product only.  The
		// mask of a key tile is drawn right before that tile is packed (drawn for all tiles up front, the hash
		// temporaries of 28 unrolled draws spill)
		uint32_t rk = 0;
		if constexpr (DROP) rk = attn_row_key(attn_head_key(a, bh), q);
		f32x4 o[4];
#pragma unroll
		for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
		for (int u = 0; u < KT2; ++u) {
			if constexpr (DROP) {
#pragma unroll
				for (int t = 2 * u; t < 2 * u + 2; ++t) {
					const int k0 = 16 * t + 4 * g;
					const uint32_t b0 = attn_pair_bits(rk, k0 >> 1), b1 = attn_pair_bits(rk, (k0 >> 1) + 1);
					st[t][0] *= attn_keep(b0, 0, a.drop_thr16, 1.f);
					st[t][1] *= attn_keep(b0, 1, a.drop_thr16, 1.f);
					st[t][2] *= attn_keep(b1, 0, a.drop_thr16, 1.f);
					st[t][3] *= attn_keep(b1, 1, a.drop_thr16, 1.f);
				}
			}
			const bf16x8 pf = pack8(st[2 * u], st[2 * u + 1]);
#pragma unroll
			for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(tr_frag(Vs, 32 * u, 32 * u + 16, 16 * dt, lane), pf, o[dt]);
			__builtin_amdgcn_sched_barrier(0);
		}
		if (q < a.Sq) {
			const float inv = (DROP ? a.drop_scale : 1.f) / l;
			bf16_t* op = reinterpret_cast<bf16_t*>(a.out_w) + ((int64_t)b * a.Sq + q) * a.ldo + h * DH + 4 * g;
#pragma unroll
			for (int dt = 0; dt < 4; ++dt) {
				bf16x4 ov;
#pragma unroll
				for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(o[dt][r] * inv);
				*reinterpret_cast<bf16x4*>(op + 16 * dt) = ov;
			}
			if (g == 0) a.lse[(int64_t)bh * a.Sq + q] = m * a.scale + __logf(l);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// backward: dQ (query on the lane)
// ---------------------------------------------------------------------------------------------------------------
// A workgroup walks `a.heads_per_wg` consecutive (batch, head) pairs like the forward kernel: two buffers of a K and a V
// image filled by LDS-DMA (the next pair's images are in flight while this pair is worked on), one 16-query tile per wave,
// whose rows of Q, dO and O for the next pair are already in registers.
template <int KT2, bool DROP>
__global__ __launch_bounds__(1024) void attn_bwd_dq_mfma_kernel(AttnArgs a) {
	if constexpr (DROP) attn_enter(a);
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr int IMG = KT2 * 32 * 128;  // one image
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
	const int g = lane >> 4;
	const int n_bh = a.B * a.n_head;
	const int bh0 = blockIdx.x * a.heads_per_wg;
	const int bh1 = min(n_bh, bh0 + a.heads_per_wg);
	const float c = a.scale * LOG2E;
	const int n_qtiles = (a.Sq + 15) >> 4;  // <= number of waves: one query tile per wave and pair
	const bool active = wave < n_qtiles;
	const int q = wave * 16 + (lane & 15);
	const int qc = q < a.Sq ? q : a.Sq - 1;

	auto stage = [&](int bh, int buf) {
		const int b = bh / a.n_head, h = bh - b * a.n_head;
		const bf16_t* kb = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Skv * a.ldk + h * DH;
		const bf16_t* vb = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Skv * a.ldv + h * DH;
		stage_image_dma(smem + buf * 2 * IMG, kb, a.ldk, a.Skv, KT2 * 32, wave, nwaves, lane);
		stage_image_dma(smem + buf * 2 * IMG + IMG, vb, a.ldv, a.Skv, KT2 * 32, wave, nwaves, lane);
	};
	bf16x8 qn0 = {}, qn1 = {}, gn0 = {}, gn1 = {}, on0 = {}, on1 = {};
	float lsen = 0.f;
	auto fetch_rows = [&](int bh) {
		const int b = bh / a.n_head, h = bh - b * a.n_head;
		const int64_t qrow = (int64_t)b * a.Sq + qc;
		const bf16_t* qp = reinterpret_cast<const bf16_t*>(a.q) + qrow * a.ldq + h * DH + 8 * g;
		const bf16_t* gp = reinterpret_cast<const bf16_t*>(a.dout) + qrow * a.ldo + h * DH + 8 * g;
		const bf16_t* op = reinterpret_cast<const bf16_t*>(a.out) + qrow * a.ldo + h * DH + 8 * g;
		qn0 = *reinterpret_cast<const bf16x8*>(qp), qn1 = *reinterpret_cast<const bf16x8*>(qp + 32);
		gn0 = *reinterpret_cast<const bf16x8*>(gp), gn1 = *reinterpret_cast<const bf16x8*>(gp + 32);
		on0 = *reinterpret_cast<const bf16x8*>(op), on1 = *reinterpret_cast<const bf16x8*>(op + 32);
		lsen = a.lse[(int64_t)bh * a.Sq + qc];
	};
	if (bh0 >= bh1) return;  // (grids never exceed the pair count)
	if (active) fetch_rows(bh0);
	stage(bh0, 0);
	for (int bh = bh0; bh < bh1; ++bh) {
		const int buf = (bh - bh0) & 1;
		// this pair's images have landed (the barrier's fence drains the DMA, and with it the rows requested before it) and
		// every wave is done with the other buffer
		__syncthreads();
		const bf16x8 qf0 = qn0, qf1 = qn1, gf0 = gn0, gf1 = gn1, of0 = on0, of1 = on1;
		const float lse2 = lsen * LOG2E;
		if (bh + 1 < bh1) {
			stage(bh + 1, buf ^ 1);
			if (active) fetch_rows(bh + 1);
		}
		if (!active) continue;
		const char* Ks = smem + buf * 2 * IMG;
		const char* Vs = Ks + IMG;
		const int b = bh / a.n_head, h = bh - b * a.n_head;
		float delta = 0.f;
#pragma unroll
		for (int j = 0; j < 8; ++j) delta += (float)gf0[j] * (float)of0[j] + (float)gf1[j] * (float)of1[j];
		delta = group_sum(delta);
		if (g == 0 && q < a.Sq) {  // row statistics for the dK/dV kernel, which runs after this one
			a.bwd_scratch[(int64_t)bh * a.Sq + q] = lse2;
			a.bwd_scratch[((int64_t)n_bh + bh) * a.Sq + q] = delta;
		}
		const uint32_t rk = DROP ? attn_row_key(attn_head_key(a, bh), q) : 0u;
		f32x4 dq[4];
#pragma unroll
		for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
		for (int u = 0; u < KT2; ++u) {
			f32x4 ds[2];
#pragma unroll
			for (int hf = 0; hf < 2; ++hf) {
				const int t = 2 * u + hf;
				f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
				s = mfma16(row_frag(Ks, 16 * t, 0, lane), qf0, s);
				s = mfma16(row_frag(Ks, 16 * t, 1, lane), qf1, s);
				dp = mfma16(row_frag(Vs, 16 * t, 0, lane), gf0, dp);
				dp = mfma16(row_frag(Vs, 16 * t, 1, lane), gf1, dp);
				if constexpr (DROP) {
					// out = (P o M) V with M in {0, 1/(1-p)}: dP = M o (dO V^T); delta = rowsum(dO o out) as without dropout
					const int k0 = 16 * t + 4 * g;
					const uint32_t b0 = attn_pair_bits(rk, k0 >> 1), b1 = attn_pair_bits(rk, (k0 >> 1) + 1);
					dp[0] *= attn_keep(b0, 0, a.drop_thr16, a.drop_scale);
					dp[1] *= attn_keep(b0, 1, a.drop_thr16, a.drop_scale);
					dp[2] *= attn_keep(b1, 0, a.drop_thr16, a.drop_scale);
					dp[3] *= attn_keep(b1, 1, a.drop_thr16, a.drop_scale);
				}
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const float p = __builtin_amdgcn_exp2f(s[r] * c - lse2);
					ds[hf][r] = p * (dp[r] - delta) * a.scale;
				}
				if (16 * t + 16 > a.Skv) {  // wave-uniform: padded keys contribute nothing
#pragma unroll
					for (int r = 0; r < 4; ++r)
						if (16 * t + 4 * g + r >= a.Skv) ds[hf][r] = 0.f;
				}
			}
			const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
			for (int dt = 0; dt < 4; ++dt) dq[dt] = mfma16(tr_frag(Ks, 32 * u, 32 * u + 16, 16 * dt, lane), dsf, dq[dt]);
		}
		if (q < a.Sq) {
			bf16_t* dqp = reinterpret_cast<bf16_t*>(a.dq) + ((int64_t)b * a.Sq + q) * a.lddq + h * DH + 4 * g;
#pragma unroll
			for (int dt = 0; dt < 4; ++dt) {
				bf16x4 ov;
#pragma unroll
				for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)dq[dt][r];
				*reinterpret_cast<bf16x4*>(dqp + 16 * dt) = ov;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// backward: dK, dV (key on the lane)
// ---------------------------------------------------------------------------------------------------------------
// Transposing LDS reads as inline asm that is issued but not waited for (lgkm_wait retires them).  Two reasons: hipcc
// puts s_waitcnt vmcnt(0) in front of the builtin whenever an LDS-DMA request is in flight (it would wait for the NEXT
// pair's images), and it schedules every builtin read right in front of its MFMA, one LDS round trip per MFMA.
// `addr`: LDS byte address of the lane's fragment in step 0 of the first image; OFF: constant byte offset (image).
template <int OFF>
__device__ __forceinline__ void tr_issue(uint32_t addr, s16x4& lo, s16x4& hi) {
	asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF));
	asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF + 2048));
}
// waits until every LDS read issued so far has returned; names the registers so that no use can move above the wait
__device__ __forceinline__ void lgkm_wait(s16x4 (&f)[4][2]) {
	asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1]));
}
__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
	union {
		s16x4 h[2];
		bf16x8 v;
	} u;
	u.h[0] = lo;
	u.h[1] = hi;
	return u.v;
}
__device__ __forceinline__ uint32_t lds_addr(const char* p) {
	typedef __attribute__((address_space(3))) void* lds_p;
	return (uint32_t)(uintptr_t)(lds_p)(void*)p;
}

// Eight waves per workgroup (two per SIMD, so a lane may use 256 registers), each wave owns up to two 16-key tiles and
// sums over all queries; a workgroup walks consecutive (batch, head) pairs.  Two buffers of {Q image, dO image, lse[q]
// (log2 units), delta[q] = sum_d dO O} are filled by LDS-DMA, the last two from the scratch the dQ kernel wrote; the
// next pair's K / V rows are already in registers.  Per 32 queries: (A) all eight row fragments are requested, (B) the S
// and dP MFMAs run, (C) the eight transposed fragments for the dV / dK products are requested, (D) the softmax
// arithmetic covers their latency, (E) the dV / dK MFMAs: two exposed LDS round trips per step instead of the fourteen
// of a one-read-per-MFMA schedule.  (The same form with one tile per wave and 13 waves has 128 registers per lane,
// spills, and is slower than the plain kernel it replaced: 176 -> 137 us per ViT-B/16 layer needs the registers.)
template <int QT2, bool DROP>  // queries padded to 32 * QT2
__global__ __launch_bounds__(512) void attn_bwd_dkv_mfma_kernel(AttnArgs a) {
	if constexpr (DROP) attn_enter(a);
	extern __shared__ __attribute__((aligned(16))) char smem[];
	constexpr int QP = QT2 * 32, IMG = QP * 128;
	constexpr int QF = (QP + 63) & ~63;
	constexpr int BUF = 2 * IMG + 2 * QF * 4;
	uint32_t* rk_s = reinterpret_cast<uint32_t*>(smem + 2 * BUF);
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	constexpr int NW = 8;
	const int g = lane >> 4;
	const int n_bh = a.B * a.n_head;
	const int bh0 = blockIdx.x * a.heads_per_wg;
	const int bh1 = min(n_bh, bh0 + a.heads_per_wg);
	const float c = a.scale * LOG2E;
	const int n_ktiles = (a.Skv + 15) >> 4;  // <= 16: two tile slots per wave

	auto stage = [&](int bh, int buf) {
		const int b = bh / a.n_head, h = bh - b * a.n_head;
		char* base = smem + buf * BUF;
		const bf16_t* qb = reinterpret_cast<const bf16_t*>(a.q) + (int64_t)b * a.Sq * a.ldq + h * DH;
		const bf16_t* gb = reinterpret_cast<const bf16_t*>(a.dout) + (int64_t)b * a.Sq * a.ldo + h * DH;
		stage_image_dma(base, qb, a.ldq, a.Sq, QP, wave, NW, lane);
		stage_image_dma(base + IMG, gb, a.ldo, a.Sq, QP, wave, NW, lane);
		for (int piece = wave; piece < 2 * (QF / 64); piece += NW) {
			const int which = piece / (QF / 64), pp = piece - which * (QF / 64);
			const int qi = 64 * pp + lane;
			const float* src = a.bwd_scratch + ((int64_t)which * n_bh + bh) * a.Sq + qi;
			if (qi < a.Sq) __builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(base + 2 * IMG + (which * QF + 64 * pp) * 4), 4, 0, 0);
		}
	};
	bf16x8 nk[2][2], nv[2][2];
#pragma unroll
	for (int j = 0; j < 2; ++j)
#pragma unroll
		for (int s2 = 0; s2 < 2; ++s2) nk[j][s2] = nv[j][s2] = bf16x8{};
	auto fetch_rows = [&](int bh) {
		const int b = bh / a.n_head, h = bh - b * a.n_head;
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			if (wave + NW * j < n_ktiles) {
				const int key = (wave + NW * j) * 16 + (lane & 15);
				const int64_t krow = (int64_t)b * a.Skv + min(key, a.Skv - 1);
				const bf16_t* kp = reinterpret_cast<const bf16_t*>(a.k) + krow * a.ldk + h * DH + 8 * g;
				const bf16_t* vp = reinterpret_cast<const bf16_t*>(a.v) + krow * a.ldv + h * DH + 8 * g;
				nk[j][0] = *reinterpret_cast<const bf16x8*>(kp), nk[j][1] = *reinterpret_cast<const bf16x8*>(kp + 32);
				nv[j][0] = *reinterpret_cast<const bf16x8*>(vp), nv[j][1] = *reinterpret_cast<const bf16x8*>(vp + 32);
			}
		}
	};
	auto row_keys = [&](int bh, int buf) {
		if constexpr (DROP) {
			const uint32_t hk = attn_head_key(a, bh);
			for (int r = threadIdx.x; r < QP; r += blockDim.x) rk_s[buf * QP + r] = attn_row_key(hk, r);
		}
	};
	// queries past the last one: lse = +inf makes their probabilities exactly zero (the DMA never writes these slots)
	for (int i = threadIdx.x; i < 2 * 2 * QF; i += blockDim.x) {
		const int bi = i / (2 * QF), r = i - bi * 2 * QF;
		float* arr = reinterpret_cast<float*>(smem + bi * BUF + 2 * IMG);
		if ((r < QF ? r : r - QF) >= a.Sq) arr[r] = r < QF ? INFINITY : 0.f;
	}
	// byte offsets of a lane's fragment reads (the tile / step part is added as an immediate)
	int rf[2];
	uint32_t trb[4];
#pragma unroll
	for (int s2 = 0; s2 < 2; ++s2) rf[s2] = img_off(lane & 15, 4 * s2 + g);
	{
		const int i = lane & 15, q4 = i >> 2, p = i & 3;
#pragma unroll
		for (int dt = 0; dt < 4; ++dt) trb[dt] = img_off(4 * g + q4, 2 * dt + (p >> 1)) + ((p & 1) << 3);
	}
	fetch_rows(bh0);
	stage(bh0, 0);
	row_keys(bh0, 0);
	for (int bh = bh0; bh < bh1; ++bh) {
		const int buf = (bh - bh0) & 1;
		__syncthreads();
		bf16x8 ck[2][2], cv[2][2];
#pragma unroll
		for (int j = 0; j < 2; ++j)
#pragma unroll
			for (int s2 = 0; s2 < 2; ++s2) {
				ck[j][s2] = nk[j][s2];
				cv[j][s2] = nv[j][s2];
			}
		if (bh + 1 < bh1) {
			stage(bh + 1, buf ^ 1);
			row_keys(bh + 1, buf ^ 1);
			fetch_rows(bh + 1);
		}
		const char* Qs = smem + buf * BUF;
		const char* Gs = Qs + IMG;
		const float* lse_s = reinterpret_cast<const float*>(Gs + IMG);
		const float* del_s = lse_s + QF;
		const int b = bh / a.n_head, h = bh - b * a.n_head;
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			if (wave + NW * j >= n_ktiles) continue;
			const int key = (wave + NW * j) * 16 + (lane & 15);
			f32x4 dk[4], dv[4];
#pragma unroll
			for (int dt = 0; dt < 4; ++dt) {
				dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
				dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
			}
			uint32_t tra[4];
#pragma unroll
			for (int dt = 0; dt < 4; ++dt) tra[dt] = lds_addr(Qs) + trb[dt];
			const char* rq = Qs + rf[0];
			const char* rq1 = Qs + rf[1];
#pragma unroll 1
			for (int u = 0; u < QT2; ++u) {
				f32x4 p[2], ds[2], sv[2], dpv[2];
				bf16x8 fq[2][2], fg[2][2];
#pragma unroll
				for (int hf = 0; hf < 2; ++hf) {
					fq[hf][0] = *reinterpret_cast<const bf16x8*>(rq + 4096 * u + 2048 * hf);
					fq[hf][1] = *reinterpret_cast<const bf16x8*>(rq1 + 4096 * u + 2048 * hf);
					fg[hf][0] = *reinterpret_cast<const bf16x8*>(rq + IMG + 4096 * u + 2048 * hf);
					fg[hf][1] = *reinterpret_cast<const bf16x8*>(rq1 + IMG + 4096 * u + 2048 * hf);
				}
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int hf = 0; hf < 2; ++hf) {
					sv[hf] = mfma16(fq[hf][0], ck[j][0], (f32x4){0.f, 0.f, 0.f, 0.f});
					dpv[hf] = mfma16(fg[hf][0], cv[j][0], (f32x4){0.f, 0.f, 0.f, 0.f});
				}
#pragma unroll
				for (int hf = 0; hf < 2; ++hf) {
					sv[hf] = mfma16(fq[hf][1], ck[j][1], sv[hf]);
					dpv[hf] = mfma16(fg[hf][1], cv[j][1], dpv[hf]);
				}
				__builtin_amdgcn_sched_barrier(0);
				s16x4 gt[4][2], qt[4][2];
#pragma unroll
				for (int dt = 0; dt < 4; ++dt) {
					tr_issue<IMG>(tra[dt], gt[dt][0], gt[dt][1]);
					tr_issue<0>(tra[dt], qt[dt][0], qt[dt][1]);
					tra[dt] += 4096;
				}
#pragma unroll
				for (int hf = 0; hf < 2; ++hf) {
					const int t = 2 * u + hf;
					const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + 16 * t + 4 * g);
					const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + 16 * t + 4 * g);
					float keep[4] = {1.f, 1.f, 1.f, 1.f};
					if constexpr (DROP) {
						const uint4 rk4 = *reinterpret_cast<const uint4*>(rk_s + buf * QP + 16 * t + 4 * g);
						keep[0] = attn_keep(attn_pair_bits(rk4.x, key >> 1), key, a.drop_thr16, a.drop_scale);
						keep[1] = attn_keep(attn_pair_bits(rk4.y, key >> 1), key, a.drop_thr16, a.drop_scale);
						keep[2] = attn_keep(attn_pair_bits(rk4.z, key >> 1), key, a.drop_thr16, a.drop_scale);
						keep[3] = attn_keep(attn_pair_bits(rk4.w, key >> 1), key, a.drop_thr16, a.drop_scale);
					}
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						const float pr = __builtin_amdgcn_exp2f(sv[hf][r] * c - l4[r]);
						p[hf][r] = DROP ? pr * keep[r] : pr;
						ds[hf][r] = pr * ((DROP ? dpv[hf][r] * keep[r] : dpv[hf][r]) - d4[r]) * a.scale;
					}
				}
				const bf16x8 pf = pack8(p[0], p[1]);
				const bf16x8 dsf = pack8(ds[0], ds[1]);
				lgkm_wait(gt);
				lgkm_wait(qt);
#pragma unroll
				for (int dt = 0; dt < 4; ++dt) {
					dv[dt] = mfma16(join8(gt[dt][0], gt[dt][1]), pf, dv[dt]);
					dk[dt] = mfma16(join8(qt[dt][0], qt[dt][1]), dsf, dk[dt]);
				}
			}
			if (key < a.Skv) {
				bf16_t* dkp = reinterpret_cast<bf16_t*>(a.dk) + ((int64_t)b * a.Skv + key) * a.lddk + h * DH + 4 * g;
				bf16_t* dvp = reinterpret_cast<bf16_t*>(a.dv) + ((int64_t)b * a.Skv + key) * a.lddv + h * DH + 4 * g;
#pragma unroll
				for (int dt = 0; dt < 4; ++dt) {
					bf16x4 kv, vv;
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						kv[r] = (bf16_t)dk[dt][r];
						vv[r] = (bf16_t)dv[dt][r];
					}
					*reinterpret_cast<bf16x4*>(dkp + 16 * dt) = kv;
					*reinterpret_cast<bf16x4*>(dvp + 16 * dt) = vv;
				}
			}
		}
	}
}


// consecutive (batch, head) pairs per workgroup of the forward kernel: as many as still leave every CU a workgroup
int heads_per_workgroup(int n_bh) {
	static int cus = 0;
	if (cus == 0) {
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
	}
	return std::max(1, std::min(16, n_bh / cus));
}

template <int T2, bool DROP>
int launch_fwd(const AttnArgs& a0, hipStream_t st) {
	AttnArgs a = a0;
	const int nw = (a.Sq + 15) / 16;  // one wave per 16-query tile (<= 16: the kernels cover up to 256 queries)
	const int n_bh = a.B * a.n_head;
	a.heads_per_wg = heads_per_workgroup(n_bh);
	const size_t lds = (size_t)4 * T2 * 32 * 128;  // two buffers of a K and a V image
	static bool attr = false;
	if (!attr) {
		hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_mfma_kernel<T2, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr = true;
	}
	hipLaunchKernelGGL((attn_fwd_mfma_kernel<T2, DROP>), dim3((unsigned)vitk_cdiv(n_bh, a.heads_per_wg)), dim3(64 * nw), lds, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}
template <int T2, bool DROP>
int launch_dq(const AttnArgs& a0, hipStream_t st) {
	AttnArgs a = a0;
	const int nw = std::min(16, (a.Sq + 15) / 16);
	const int n_bh = a.B * a.n_head;
	a.heads_per_wg = heads_per_workgroup(n_bh);
	const size_t lds = (size_t)4 * T2 * 32 * 128;  // two buffers of a K and a V image
	static bool attr = false;
	if (!attr) {
		hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dq_mfma_kernel<T2, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr = true;
	}
	hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<T2, DROP>), dim3((unsigned)vitk_cdiv(n_bh, a.heads_per_wg)), dim3(64 * nw), lds, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}
template <int T2, bool DROP>
int launch_dkv(const AttnArgs& a0, hipStream_t st) {
	AttnArgs a = a0;
	const int n_bh = a.B * a.n_head;
	a.heads_per_wg = heads_per_workgroup(n_bh);
	const int qp = T2 * 32, qf = (qp + 63) & ~63;
	const size_t lds = (size_t)2 * (2 * qp * 128 + 2 * qf * 4) + (size_t)2 * qp * sizeof(uint32_t);
	static bool attr = false;
	if (!attr) {
		hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dkv_mfma_kernel<T2, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr = true;
	}
	hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<T2, DROP>), dim3((unsigned)vitk_cdiv(n_bh, a.heads_per_wg)), dim3(512), lds, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

#define DISPATCH_CASE(n, FN) \
	case n: return a.drop_thr16 != 0 ? FN<n, true>(a, st) : FN<n, false>(a, st);
#define DISPATCH_T2(n, FN)                    \
	switch (n) {                              \
		DISPATCH_CASE(1, FN)                  \
		DISPATCH_CASE(2, FN)                  \
		DISPATCH_CASE(3, FN)                  \
		DISPATCH_CASE(4, FN)                  \
		DISPATCH_CASE(5, FN)                  \
		DISPATCH_CASE(6, FN)                  \
		DISPATCH_CASE(7, FN)                  \
		DISPATCH_CASE(8, FN)                  \
		default: return VITK_ERR_UNSUPPORTED; \
	}

int run_fwd(const AttnArgs& a, hipStream_t st) { DISPATCH_T2((a.Skv + 31) / 32, launch_fwd) }
int run_dq(const AttnArgs& a, hipStream_t st) { DISPATCH_T2((a.Skv + 31) / 32, launch_dq) }
int run_dkv(const AttnArgs& a, hipStream_t st) { DISPATCH_T2((a.Sq + 31) / 32, launch_dkv) }

bool al16(const void* p) { return p == nullptr || vitk_aligned(p, 16); }
bool al8(const void* p) { return p == nullptr || vitk_aligned(p, 8); }

}  // namespace

bool vitk_attention_mfma_ok(const AttnArgs& a, int dtype) {
	if (dtype != VITK_BF16 || a.dh != DH) return false;
	if (a.Skv > 256 || a.Sq > 256) return false;  // the dK/dV kernel keeps all queries in LDS, the others all keys
	if ((a.ldq | a.ldk | a.ldv | a.ldo) % 8 != 0) return false;
	if (a.dq != nullptr && ((a.lddq | a.lddk | a.lddv) % 4 != 0)) return false;
	return al16(a.q) && al16(a.k) && al16(a.v) && al16(a.out) && al16(a.dout) && al8(a.out_w) && al8(a.dq) && al8(a.dk) && al8(a.dv);
}

int vitk_attention_mfma(const AttnArgs& a, bool bwd, hipStream_t st) {
	if (!bwd) return run_fwd(a, st);
	if (a.bwd_scratch == nullptr) return VITK_ERR_NULL;
	if (vitk_attention_bwd_fused_ok(a)) return vitk_attention_bwd_fused(a, st);  // one pass: S and dP formed once
	int rc = run_dq(a, st);  // also leaves lse (log2 units) and delta per query row in the scratch
	if (rc != VITK_OK) return rc;
	return run_dkv(a, st);
}
