// Backward of the short-sequence bf16 attention (head size 64) in ONE pass: autograd of F.scaled_dot_product_attention,
// layers/attention.py:39-43 of the reference.  S = Q K^T and dP = dO V^T are formed once per (batch, head) pair and all
// three gradients come out of them -- five matrix products instead of the seven of the two-kernel form (one kernel with the
// query on the MFMA lane for dQ, one with the key on the lane for dK / dV, each recomputing S and dP and the softmax
// arithmetic), every operand read from HBM once, no per-row statistics passed between kernels through HBM.
//
// One 512-thread workgroup walks consecutive (batch, head) pairs.  A wave owns up to two 16-key tiles (tiles w and w + 8)
// whose K and V rows it holds in registers, and the queries go by in STEPS of 32:
//   region 1 (key on the lane)   S, dP for the wave's keys x 32 queries; delta = rowsum(dO o O) as the diagonal of dO O^T
//                                (two more MFMAs per 16 queries, no cross-wave exchange); P, dS; dV^T += dO^T P,
//                                dK^T += Q^T dS; the dS block goes to an LDS image [key][query]
//   barrier
//   region 2 (query on the lane) each wave forms one 16 x 16 tile of dQ^T = K^T dS^T for those 32 queries over ALL keys
//                                (transposing reads of the K image and of the dS image) and stores it
// One barrier per step: the dS image holds two steps side by side (the halves of its 128-byte rows), so a wave may write
// the next step's block while others still read this one's.
//
// Everything that comes from HBM arrives by LDS-DMA (global_load_lds_dwordx4) and is never waited for with vmcnt(0) in
// steady state: a RING of three slots holds the Q, dO, O rows (+ lse) of a step and is refilled two steps ahead, across pair
// boundaries too; the K images of two consecutive pairs alternate between two buffers, the V image (read once per pair, into
// registers) has one.  Each wave issues a fixed number of requests per step, so "my requests for step s + 1 have landed" is a
// counted s_waitcnt vmcnt(N) in front of the step's barrier, N = the number of younger operations (the dQ store, the next
// pair's K / V requests, the dK / dV stores), each of which is issued unconditionally (stores of rows that do not exist go to
// the caller's scratch) so that the count is exact.  All LDS traffic is inline asm with explicit lgkmcnt waits: in front of a
// compiler-visible LDS access that may alias an LDS-DMA in flight hipcc emits s_waitcnt vmcnt(0).
#include "attention.h"

#include <atomic>
#include <cstdlib>

namespace {

constexpr int DH = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int SLOT = 3 * 4096 + 256;  // Q, dO, O images of 32 rows + 64 floats of lse (32 used)
constexpr int RING = 3;

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(lds_void_p)(void*)p; }
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// LDS accesses that are issued here and retired by an explicit wait naming their registers
__device__ __forceinline__ void rd128(bf16x8& dst, uint32_t addr, int offset) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(offset)); }
__device__ __forceinline__ void rd128f(f32x4& dst, uint32_t addr, int offset) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(offset)); }
__device__ __forceinline__ void rdtr(s16x4& dst, uint32_t addr, int offset) { asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(offset)); }
__device__ __forceinline__ void wr64(uint32_t addr, s16x4 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
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

// One 8-row piece (1 KiB, one wave-instruction) of a [rows][64] bf16 matrix into image rows [8 piece, 8 piece + 8): written
// linearly at a wave-uniform LDS address, the image's chunk swizzle (16-byte chunk ^ (row & 7)) applied to each lane's SOURCE
// address.  Rows past rows_valid re-read the last valid row (finite values that only ever meet zero probabilities).
__device__ __forceinline__ void dma_piece(char* img, const bf16_t* src, int64_t ld, int rows_valid, int piece, int lane) {
	// (an opaque copy of the lane index per request: addresses formed from a loop-invariant lane index are hoisted out of the
	// pair loop, eight 64-bit values for the K / V pieces alone, spilled, and every reload is followed by s_waitcnt vmcnt(0))
	asm volatile("" : "+v"(lane));
	const int r8 = lane >> 3, sl = lane & 7;
	const int row = 8 * piece + r8;
	const int rs = row < rows_valid ? row : rows_valid - 1;
	const bf16_t* p = src + (int64_t)rs * ld + 8 * (sl ^ (row & 7));
	__builtin_amdgcn_global_load_lds((gbl_void_p)p, (lds_void_p)(img + piece * 1024), 16, 0, 0);
}

template <int KT2, bool DROP>  // keys padded to 32 * KT2 (KT2 <= 7)
__global__ __launch_bounds__(512) void attn_bwd_fused_kernel(AttnArgs a) {
	if constexpr (DROP) attn_enter(a);
	extern __shared__ __attribute__((aligned(128))) char smem[];
	constexpr int KIMG = KT2 * 32 * 128;
	constexpr int KOFF = RING * SLOT, VOFF = KOFF + 2 * KIMG, DSOFF = VOFF + KIMG;
	constexpr int KVPW = (4 * KT2 + 7) / 8;  // requests per wave for one K or V image
	constexpr int NKV = 2 * KVPW;
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	// (an opaque copy of the lane index: lane-dependent addresses formed from the original are hoisted, spilled and reloaded)
	int lane = tid & 63;
	asm volatile("" : "+v"(lane));
	const int g = lane >> 4, i16 = lane & 15;
	const int n_bh = a.B * a.n_head;
	const int bh0 = blockIdx.x * a.heads_per_wg;
	const int npairs = min(n_bh, bh0 + a.heads_per_wg) - bh0;
	if (npairs <= 0) return;  // (grids never exceed the pair count)
	const int QT2 = (a.Sq + 31) >> 5;  // >= 2 (the host sends shorter query sets to the two-kernel form)
	const int n_ktiles = (a.Skv + 15) >> 4;
	const bool own0 = wave < n_ktiles, own1 = wave + 8 < n_ktiles;
	const int nst = 8 * ((own0 ? 1 : 0) + (own1 ? 1 : 0));  // dK / dV stores of this wave at the end of a pair
	const float c = a.scale * LOG2E;
	const uint32_t lds0 = lds_addr(smem);
	char* const dump = reinterpret_cast<char*>(a.bwd_scratch);  // where stores of rows that do not exist go

	// ---- request issue --------------------------------------------------------------------------------------------------
	// the Q, dO, O rows and lse of query step u of pair pi into ring slot `slot`: exactly two requests per wave (16 in all:
	// 4 pieces per image, one of lse, and three repeats of O pieces that keep the count uniform)
	auto issue_ring = [&](int pi, int u, int slot) {
		const int bh = bh0 + pi;
		const int b = bh / a.n_head, h = bh - b * a.n_head;
		char* sp = smem + slot * SLOT;
		const int rows_valid = a.Sq - 32 * u;
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const int o = 2 * wave + j;
			if (o == 12) {
				int ln = lane;
				asm volatile("" : "+v"(ln));
				const int qi = min(32 * u + ln, a.Sq - 1);
				const float* src = a.lse + (int64_t)bh * a.Sq + qi;
				__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(sp + 3 * 4096), 4, 0, 0);
			} else {
				const int x = o < 12 ? (o >> 2) : 2, pc = o < 12 ? (o & 3) : (o - 13);
				const bf16_t* base = reinterpret_cast<const bf16_t*>(x == 0 ? a.q : (x == 1 ? a.dout : a.out));
				const int64_t ld = x == 0 ? a.ldq : a.ldo;
				dma_piece(sp + x * 4096, base + ((int64_t)b * a.Sq + 32 * u) * ld + h * DH, ld, rows_valid, pc, lane);
			}
		}
	};
	// the K image of pair pi into K buffer kb and its V image: NKV requests per wave (pieces past the last repeat it)
	auto issue_kv = [&](int pi, int kb) {
		const int bh = bh0 + pi;
		const int b = bh / a.n_head, h = bh - b * a.n_head;
		const bf16_t* kbase = reinterpret_cast<const bf16_t*>(a.k) + (int64_t)b * a.Skv * a.ldk + h * DH;
		const bf16_t* vbase = reinterpret_cast<const bf16_t*>(a.v) + (int64_t)b * a.Skv * a.ldv + h * DH;
#pragma unroll
		for (int t = 0; t < KVPW; ++t) dma_piece(smem + KOFF + kb * KIMG, kbase, a.ldk, a.Skv, min(wave + 8 * t, 4 * KT2 - 1), lane);
#pragma unroll
		for (int t = 0; t < KVPW; ++t) dma_piece(smem + VOFF, vbase, a.ldv, a.Skv, min(wave + 8 * t, 4 * KT2 - 1), lane);
	};
	auto step_of = [&](int s, int& pi, int& u) {  // global step -> (pair, query step); past the end: the last step again
		const int total = npairs * QT2;
		const int sc = s < total ? s : total - 1;
		pi = sc / QT2;
		u = sc - pi * QT2;
	};

	// ---- lane offsets ---------------------------------------------------------------------------------------------------
	// row fragment of a [*][64] image: row (i16) of a 16-row block, 16-byte chunk 4 s2 + g, swizzled; s2 = 1 flips bit 6
	const uint32_t rf0 = (uint32_t)(i16 * 128 + ((g ^ (i16 & 7)) << 4));
	const uint32_t rf1 = rf0 ^ 64u;
	// transposing reads: lane (g, q4, p) = (lane >> 4, i16 >> 2, i16 & 3) addresses row 4 g + q4 (+ 16 for the second read) and
	// 8 bytes at column 16 dt + 4 p of a 16-column block
	const int q4 = i16 >> 2, pp = i16 & 3;
	const int trow = 4 * g + q4;
	auto tr_off = [&](int chunk2) { return (uint32_t)(trow * 128 + (((chunk2 + (pp >> 1)) ^ (trow & 7)) << 4) + ((pp & 1) << 3)); };
	uint32_t trb[4];
#pragma unroll
	for (int dt = 0; dt < 4; ++dt) trb[dt] = tr_off(2 * dt);
	// region 2: this wave's tile of dQ^T
	const int qtw = wave >> 2, dtw = wave & 3;
	const uint32_t trk = tr_off(2 * dtw);
	const uint32_t trs[2] = {tr_off(2 * qtw), tr_off(4 + 2 * qtw)};  // dS image: step parity 0 / 1
	// dS block of a key tile: row i16 of the tile, the lane's four queries 16 hf + 4 g .. + 3 are 8 bytes at logical chunk
	// 4 parity + 2 hf + (g >> 1), half (g & 1)
	uint32_t dsw[2];
#pragma unroll
	for (int hf = 0; hf < 2; ++hf) dsw[hf] = (uint32_t)(i16 * 128 + (((2 * hf + (g >> 1)) ^ (i16 & 7)) << 4) + ((g & 1) << 3));

	// ---- prologue -------------------------------------------------------------------------------------------------------
	// the dS image starts at zero: rows of key tiles that no wave owns (up to the padded key count) are read by region 2
	for (int i = tid; i < KIMG / 16; i += 512) reinterpret_cast<uint4*>(smem + DSOFF)[i] = make_uint4(0, 0, 0, 0);
	__syncthreads();
	issue_kv(0, 0);
	{
		int pi, u;
		step_of(0, pi, u);
		issue_ring(pi, u, 0);
		step_of(1, pi, u);
		issue_ring(pi, u, 1);
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_s_barrier();

	int s = 0, slot = 0;
#pragma clang loop unroll(disable)
	for (int pi = 0; pi < npairs; ++pi) {
		const int bh = bh0 + pi;
		const int b = bh / a.n_head, h = bh - b * a.n_head;
		const int kb = pi & 1;
		const uint32_t kimg = lds0 + KOFF + kb * KIMG, vimg = lds0 + VOFF, dsimg = lds0 + DSOFF;
		const uint32_t hk = DROP ? attn_head_key(a, bh) : 0u;
		// K and V rows of this wave's key tiles (published by the barrier that ended the previous pair)
		bf16x8 ck[2][2], cv[2][2];
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const uint32_t to = (uint32_t)((wave + 8 * j < n_ktiles ? wave + 8 * j : 0) * 2048);
			rd128(ck[j][0], kimg + to + rf0, 0);
			rd128(ck[j][1], kimg + to + rf1, 0);
			rd128(cv[j][0], vimg + to + rf0, 0);
			rd128(cv[j][1], vimg + to + rf1, 0);
		}
		asm volatile("s_waitcnt lgkmcnt(0)"
					 : "+v"(ck[0][0]), "+v"(ck[0][1]), "+v"(ck[1][0]), "+v"(ck[1][1]), "+v"(cv[0][0]), "+v"(cv[0][1]), "+v"(cv[1][0]), "+v"(cv[1][1])
					 :
					 : "memory");
		f32x4 dk[2][4], dv[2][4];
#pragma unroll
		for (int j = 0; j < 2; ++j)
#pragma unroll
			for (int dt = 0; dt < 4; ++dt) {
				dk[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
				dv[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
			}
#pragma clang loop unroll(disable)
		for (int u = 0; u < QT2; ++u, ++s) {
			const uint32_t sb = lds0 + slot * SLOT;
			const int par = s & 1;
			// ================= region 1: key on the lane =================
			if (own0) {
				bf16x8 fq[2][2], fg[2][2];
				float d4[2][4], lg[2][4];
				{
					bf16x8 fo[2][2];
					f32x4 l4[2];
#pragma unroll
					for (int hf = 0; hf < 2; ++hf) {
						rd128(fg[hf][0], sb + rf0, 4096 + 2048 * hf);
						rd128(fg[hf][1], sb + rf1, 4096 + 2048 * hf);
						rd128(fo[hf][0], sb + rf0, 8192 + 2048 * hf);
						rd128(fo[hf][1], sb + rf1, 8192 + 2048 * hf);
						rd128f(l4[hf], sb + 16 * g, 3 * 4096 + 64 * hf);
					}
#pragma unroll
					for (int hf = 0; hf < 2; ++hf) {
						rd128(fq[hf][0], sb + rf0, 2048 * hf);
						rd128(fq[hf][1], sb + rf1, 2048 * hf);
					}
					// (the four Q fragments, requested last, may still be on their way)
					asm volatile("s_waitcnt lgkmcnt(4)"
								 : "+v"(fg[0][0]), "+v"(fg[0][1]), "+v"(fg[1][0]), "+v"(fg[1][1]), "+v"(fo[0][0]), "+v"(fo[0][1]), "+v"(fo[1][0]), "+v"(fo[1][1]), "+v"(l4[0]), "+v"(l4[1])
								 :
								 : "memory");
					// delta[q] = sum_d dO[q][d] O[q][d]: the diagonal of dO O^T (rows 4 g + r on this lane, column i16)
					f32x4 dd[2];
#pragma unroll
					for (int hf = 0; hf < 2; ++hf) {
						dd[hf] = mfma16(fg[hf][0], fo[hf][0], (f32x4){0.f, 0.f, 0.f, 0.f});
						dd[hf] = mfma16(fg[hf][1], fo[hf][1], dd[hf]);
					}
					asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fq[0][0]), "+v"(fq[0][1]), "+v"(fq[1][0]), "+v"(fq[1][1])::"memory");
					// the diagonal element of row 4 g + r sits in lane 16 g + 4 g + r, register r (a cross-lane move that the compiler
					// retires with lgkmcnt(0): made while no read issued here is in flight)
					const int dsrc = 20 * g;
#pragma unroll
					for (int hf = 0; hf < 2; ++hf)
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							d4[hf][r] = __shfl(dd[hf][r], dsrc + r, 64);
							lg[hf][r] = l4[hf][r] * LOG2E;
						}
					asm volatile("" : "+v"(d4[0][0]), "+v"(d4[0][1]), "+v"(d4[0][2]), "+v"(d4[0][3]), "+v"(d4[1][0]), "+v"(d4[1][1]), "+v"(d4[1][2]), "+v"(d4[1][3])::"memory");
				}
				if (u == QT2 - 1) {  // queries past the last one: lse = +inf makes their probabilities exactly zero
#pragma unroll
					for (int hf = 0; hf < 2; ++hf)
#pragma unroll
						for (int r = 0; r < 4; ++r)
							if (32 * u + 16 * hf + 4 * g + r >= a.Sq) lg[hf][r] = INFINITY;
				}
				uint32_t rk[2][4];
				if constexpr (DROP) {
#pragma unroll
					for (int hf = 0; hf < 2; ++hf)
#pragma unroll
						for (int r = 0; r < 4; ++r) rk[hf][r] = attn_row_key(hk, 32 * u + 16 * hf + 4 * g + r);
				}
				// S, dP, then P and dS of one key tile; the dS block goes to the LDS image
				bf16x8 pf[2], dsf[2];
				auto tile_softmax = [&](int j) {
					f32x4 sv[2], dpv[2];
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
					const int tile = wave + 8 * j;
					const int key = 16 * tile + i16;
					f32x4 p[2], ds[2];
#pragma unroll
					for (int hf = 0; hf < 2; ++hf)
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							const float pr = __builtin_amdgcn_exp2f(sv[hf][r] * c - lg[hf][r]);
							float keep = 1.f;
							if constexpr (DROP) keep = attn_keep(attn_pair_bits(rk[hf][r], key >> 1), key, a.drop_thr16, a.drop_scale);
							p[hf][r] = DROP ? pr * keep : pr;
							ds[hf][r] = pr * ((DROP ? dpv[hf][r] * keep : dpv[hf][r]) - d4[hf][r]) * a.scale;
						}
					if (16 * tile + 16 > a.Skv && key >= a.Skv) {  // keys past the last one (their rows repeat it) contribute nothing to dQ
#pragma unroll
						for (int hf = 0; hf < 2; ++hf) {
							p[hf] = (f32x4){0.f, 0.f, 0.f, 0.f};
							ds[hf] = (f32x4){0.f, 0.f, 0.f, 0.f};
						}
					}
					pf[j] = pack8(p[0], p[1]);
					dsf[j] = pack8(ds[0], ds[1]);
					union {
						bf16x8 v;
						s16x4 h[2];
					} w;
					w.v = dsf[j];
					const uint32_t rowb = dsimg + (uint32_t)tile * 2048u;
					wr64(rowb + (dsw[0] ^ (uint32_t)(par << 6)), w.h[0]);
					wr64(rowb + (dsw[1] ^ (uint32_t)(par << 6)), w.h[1]);
				};
				tile_softmax(0);
				__builtin_amdgcn_sched_barrier(0);
				if (own1) tile_softmax(1);
				__builtin_amdgcn_sched_barrier(0);
				// dV^T += dO^T P, dK^T += Q^T dS: the transposed fragments in two halves of the head size
#pragma unroll
				for (int half = 0; half < 2; ++half) {
					s16x4 gt[2][2], qt[2][2];
#pragma unroll
					for (int d2 = 0; d2 < 2; ++d2) {
						rdtr(qt[d2][0], sb + trb[2 * half + d2], 0);
						rdtr(qt[d2][1], sb + trb[2 * half + d2], 2048);
						rdtr(gt[d2][0], sb + trb[2 * half + d2], 4096);
						rdtr(gt[d2][1], sb + trb[2 * half + d2], 4096 + 2048);
					}
					asm volatile("s_waitcnt lgkmcnt(0)"
								 : "+v"(gt[0][0]), "+v"(gt[0][1]), "+v"(gt[1][0]), "+v"(gt[1][1]), "+v"(qt[0][0]), "+v"(qt[0][1]), "+v"(qt[1][0]), "+v"(qt[1][1])
								 :
								 : "memory");
#pragma unroll
					for (int d2 = 0; d2 < 2; ++d2) {
						const int dt = 2 * half + d2;
						const bf16x8 ag = join8(gt[d2][0], gt[d2][1]), aq = join8(qt[d2][0], qt[d2][1]);
						dv[0][dt] = mfma16(ag, pf[0], dv[0][dt]);
						dk[0][dt] = mfma16(aq, dsf[0], dk[0][dt]);
						if (own1) {
							dv[1][dt] = mfma16(ag, pf[1], dv[1][dt]);
							dk[1][dt] = mfma16(aq, dsf[1], dk[1][dt]);
						}
					}
				}
				if (u == QT2 - 1) {  // this pair's dK, dV: 8 stores per owned tile, each issued whether or not the lane's key exists
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						if (j == 1 && !own1) continue;
						int ln = lane;
						asm volatile("" : "+v"(ln));
						const int key = 16 * (wave + 8 * j) + (ln & 15);
						const bool ok = key < a.Skv;
						char* dkp = reinterpret_cast<char*>(reinterpret_cast<bf16_t*>(a.dk) + ((int64_t)b * a.Skv + key) * a.lddk + h * DH + 4 * (ln >> 4));
						char* dvp = reinterpret_cast<char*>(reinterpret_cast<bf16_t*>(a.dv) + ((int64_t)b * a.Skv + key) * a.lddv + h * DH + 4 * (ln >> 4));
#pragma unroll
						for (int dt = 0; dt < 4; ++dt) {
							bf16x4 kv, vv;
#pragma unroll
							for (int r = 0; r < 4; ++r) {
								kv[r] = (bf16_t)dk[j][dt][r];
								vv[r] = (bf16_t)dv[j][dt][r];
							}
							*reinterpret_cast<bf16x4*>(ok ? dkp + 32 * dt : dump) = kv;
							*reinterpret_cast<bf16x4*>(ok ? dvp + 32 * dt : dump) = vv;
						}
					}
				}
			}
			// this wave's dS block is in LDS and its requests for the next step have landed; everything younger stays in flight:
			// the dQ store of the previous step, the next pair's K / V requests if the previous step issued them (unless this
			// step ends the pair: then they must have landed too), and the dK / dV stores above
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			if (u == QT2 - 1) {
				if (nst == 16) asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
				else if (nst == 8) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
				else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
			} else if (u == 1) {  // (QT2 > 2 here)
				asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 + NKV) : "memory");
			} else {
				asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
			}
			__builtin_amdgcn_s_barrier();
			// ================= region 2: query on the lane =================
			{
				int npi, nu;
				step_of(s + 2, npi, nu);
				issue_ring(npi, nu, slot == 0 ? 2 : slot - 1);  // the slot of step s - 1
			}
			if (u == 0) issue_kv(min(pi + 1, npairs - 1), kb ^ 1);
			asm volatile("" ::: "memory");
			{
				const uint32_t ab = kimg + trk, bb = dsimg + trs[par];
				f32x4 dq = {0.f, 0.f, 0.f, 0.f};
				constexpr int H0 = (KT2 + 1) / 2;  // key chunks of 32 in two batches (registers)
#pragma unroll
				for (int bt = 0; bt < 2; ++bt) {
					const int c0 = bt == 0 ? 0 : H0, c1 = bt == 0 ? H0 : KT2;
					s16x4 ka[H0][2], sa[H0][2];
#pragma unroll
					for (int cc = c0; cc < c1; ++cc) {
						rdtr(ka[cc - c0][0], ab, 4096 * cc);
						rdtr(ka[cc - c0][1], ab, 4096 * cc + 2048);
						rdtr(sa[cc - c0][0], bb, 4096 * cc);
						rdtr(sa[cc - c0][1], bb, 4096 * cc + 2048);
					}
#pragma unroll
					for (int cc = c0; cc < c1; ++cc) {
						// (wait for this chunk's four reads only: those of the later chunks of the batch stay in flight)
						asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ka[cc - c0][0]), "+v"(ka[cc - c0][1]), "+v"(sa[cc - c0][0]), "+v"(sa[cc - c0][1]) : "n"(4 * (c1 - 1 - cc)) : "memory");
						dq = mfma16(join8(ka[cc - c0][0], ka[cc - c0][1]), join8(sa[cc - c0][0], sa[cc - c0][1]), dq);
					}
				}
				int ln = lane;
				asm volatile("" : "+v"(ln));
				const int q = 32 * u + 16 * qtw + (ln & 15);
				char* dqp = reinterpret_cast<char*>(reinterpret_cast<bf16_t*>(a.dq) + ((int64_t)b * a.Sq + q) * a.lddq + h * DH + 16 * dtw + 4 * (ln >> 4));
				bf16x4 ov;
#pragma unroll
				for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)dq[r];
				*reinterpret_cast<bf16x4*>(q < a.Sq ? dqp : dump) = ov;
			}
			asm volatile("" ::: "memory");
			slot = slot == RING - 1 ? 0 : slot + 1;
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (repeated) requests must not outlive the workgroup's LDS
}

int pairs_per_workgroup(int n_bh) {
	static int cus = 0;
	if (cus == 0) {
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
	}
	return std::max(1, std::min(16, n_bh / cus));
}

template <int T2, bool DROP>
int launch_fused(const AttnArgs& a0, hipStream_t st) {
	AttnArgs a = a0;
	const int n_bh = a.B * a.n_head;
	a.heads_per_wg = pairs_per_workgroup(n_bh);
	const size_t lds = (size_t)RING * SLOT + (size_t)4 * T2 * 32 * 128;
	static bool attr = false;
	if (!attr) {
		hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<T2, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr = true;
	}
	hipLaunchKernelGGL((attn_bwd_fused_kernel<T2, DROP>), dim3((unsigned)vitk_cdiv(n_bh, a.heads_per_wg)), dim3(512), lds, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

}  // namespace

// Shapes the one-pass kernel takes (beyond what vitk_attention_mfma_ok checks): the K / V / dS images of up to 224 keys and
// the ring fit in 160 KiB of LDS, and at least two query steps per pair (the next pair's K / V requests are issued behind
// the first barrier of a pair and must land before its last).
static std::atomic<int> g_one_pass{-1};  // -1: not set yet (the environment decides), 0 / 1

extern "C" int vitk_attention_bwd_one_pass(int on) {
	int prev = g_one_pass.exchange(on ? 1 : 0, std::memory_order_relaxed);
	if (prev < 0) {
		const char* e = getenv("VITK_ATTN_BWD_FUSED");
		prev = (e != nullptr && e[0] == '0') ? 0 : 1;
	}
	return prev;
}

bool vitk_attention_bwd_fused_ok(const AttnArgs& a) {
	int on = g_one_pass.load(std::memory_order_relaxed);
	if (on < 0) {
		const char* e = getenv("VITK_ATTN_BWD_FUSED");
		on = (e != nullptr && e[0] == '0') ? 0 : 1;
		g_one_pass.store(on, std::memory_order_relaxed);
	}
	return on == 1 && a.Skv <= 224 && a.Sq > 32 && a.bwd_scratch != nullptr;
}

int vitk_attention_bwd_fused(const AttnArgs& a, hipStream_t st) {
	const int t2 = (a.Skv + 31) / 32;
#define VITK_FUSED_CASE(n) \
	case n: return a.drop_thr16 != 0 ? launch_fused<n, true>(a, st) : launch_fused<n, false>(a, st);
	switch (t2) {
		VITK_FUSED_CASE(1)
		VITK_FUSED_CASE(2)
		VITK_FUSED_CASE(3)
		VITK_FUSED_CASE(4)
		VITK_FUSED_CASE(5)
		VITK_FUSED_CASE(6)
		VITK_FUSED_CASE(7)
		default: return VITK_ERR_UNSUPPORTED;
	}
#undef VITK_FUSED_CASE
}
