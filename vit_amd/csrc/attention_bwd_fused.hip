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
#ifndef VITK_ATTN_DELTA_AHEAD
#define VITK_ATTN_DELTA_AHEAD 0
#endif
#ifndef VITK_ATTN_TILES_TOGETHER
#define VITK_ATTN_TILES_TOGETHER 0
#endif
#ifndef VITK_ATTN_PRIO
#define VITK_ATTN_PRIO 0
#endif

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4* gbl_u32x4_p;
typedef __attribute__((address_space(1))) bf16x4* gbl_bf16x4_p;

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

// at most n (wave-uniform) vector-memory operations of this wave stay outstanding
__device__ __forceinline__ void vm_wait(int n) {
#define VITK_VMW(k) \
	case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
	switch (n) {
		VITK_VMW(1) VITK_VMW(2) VITK_VMW(3) VITK_VMW(4) VITK_VMW(5) VITK_VMW(6) VITK_VMW(7) VITK_VMW(8) VITK_VMW(9) VITK_VMW(10) VITK_VMW(11) VITK_VMW(12)
		VITK_VMW(13) VITK_VMW(14) VITK_VMW(15) VITK_VMW(16) VITK_VMW(17) VITK_VMW(18) VITK_VMW(19) VITK_VMW(20)
		default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
	}
#undef VITK_VMW
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

// Diagnostic build (-DVITK_ATTN_STAMPS): cycles per segment of a query step, summed over the steps of a workgroup, per wave
// (the first 64 workgroups), into the caller's scratch from float 4096 on: tools/diag/attn_stamps.py prints them.
#ifdef VITK_ATTN_STAMPS
#define STAMP(i)                                           \
	do {                                                   \
		const uint64_t t_ = __builtin_readcyclecounter();  \
		seg[i] += (uint32_t)(t_ - tprev);                  \
		tprev = t_;                                        \
	} while (0)
#else
#define STAMP(i)
#endif

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
	// Scalars of the argument block as opaque register copies: a value the compiler knows to be a kernel argument is loaded
	// again from the argument block wherever registers are short -- an s_load and a wait for it in the middle of the step
	// loop; a value it cannot trace back is kept (or moved to a lane of a vector register, which costs nothing to read).
	int Sq = a.Sq, Skv = a.Skv, n_head = a.n_head;
	int ldq = (int)a.ldq, ldk = (int)a.ldk, ldv = (int)a.ldv, ldo = (int)a.ldo, lddq = (int)a.lddq, lddk = (int)a.lddk, lddv = (int)a.lddv;
	asm volatile("" : "+s"(Sq), "+s"(Skv), "+s"(n_head), "+s"(ldq), "+s"(ldk), "+s"(ldv), "+s"(ldo), "+s"(lddq), "+s"(lddk), "+s"(lddv));
	const char *pq = (const char*)a.q, *pk = (const char*)a.k, *pv = (const char*)a.v, *pg = (const char*)a.dout, *po = (const char*)a.out, *pl = (const char*)a.lse;
	char *pdq = (char*)a.dq, *pdk = (char*)a.dk, *pdv = (char*)a.dv;
	char* dump = reinterpret_cast<char*>(a.bwd_scratch);  // where stores of rows that do not exist go
	asm volatile("" : "+s"(pq), "+s"(pk), "+s"(pv), "+s"(pg), "+s"(po), "+s"(pl), "+s"(pdq), "+s"(pdk), "+s"(pdv), "+s"(dump));
	float scale = a.scale;
	asm volatile("" : "+s"(scale));
	const int n_bh = a.B * n_head;
	const int bh0 = blockIdx.x * a.heads_per_wg;
	const int npairs = min(n_bh, bh0 + a.heads_per_wg) - bh0;
	if (npairs <= 0) return;  // (grids never exceed the pair count)
	const int QT2 = (Sq + 31) >> 5;  // >= 2 (the host sends shorter query sets to the two-kernel form)
	const int n_ktiles = (Skv + 15) >> 4;
	const bool own0 = wave < n_ktiles, own1 = wave + 8 < n_ktiles;
	const int nst = 4 * ((own0 ? 1 : 0) + (own1 ? 1 : 0));  // dK / dV stores of this wave at the end of a pair
	const float c = scale * LOG2E;
	const uint32_t lds0 = lds_addr(smem);

	// ---- request issue --------------------------------------------------------------------------------------------------
	// Ring: the Q, dO, O rows and lse of one query step are 16 requests, exactly two per wave (4 pieces of 8 rows per image,
	// one of lse, and three repeats of O pieces that keep the count uniform).  What a wave's two requests fetch never
	// changes: array, row pitch, piece and the lane's place in it are set up once.
	const char* r_arr[2];
	int r_ldb[2], r_dst[2], r_rowl[2], r_inrow[2];
	bool r_lse[2];
#pragma unroll
	for (int j = 0; j < 2; ++j) {
		const int o = 2 * wave + j;
		r_lse[j] = o == 12;
		const int x = o < 12 ? (o >> 2) : 2, pc = o < 12 ? (o & 3) : (o - 13);
		r_arr[j] = r_lse[j] ? pl : (x == 0 ? pq : (x == 1 ? pg : po));
		r_ldb[j] = r_lse[j] ? 4 : 2 * (x == 0 ? ldq : ldo);
		r_dst[j] = r_lse[j] ? 3 * 4096 : x * 4096 + pc * 1024;
		r_rowl[j] = r_lse[j] ? lane : 8 * pc + (lane >> 3);
		r_inrow[j] = r_lse[j] ? 0 : (((lane & 7) ^ (r_rowl[j] & 7)) << 4);  // the image's chunk swizzle, applied to the source
	}
	// position of the next ring request: pair (batch ib, head ih), query step iu; past the last step it stays there (the
	// repeated requests keep every wave's count of operations in flight what the waits below assume)
	int i_pi = 0, i_u = 0, i_b = bh0 / n_head, i_h = bh0 - i_b * n_head;
	auto issue_ring = [&](int slot) {
		const int rv = Sq - 32 * i_u;  // rows of this step that exist: the others repeat the last one
		const int64_t rowI = (int64_t)i_b * Sq + 32 * i_u, rowL = ((int64_t)i_b * n_head + i_h) * Sq + 32 * i_u;
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const uint32_t voff = (uint32_t)(min(r_rowl[j], rv - 1) * r_ldb[j] + r_inrow[j]);
			char* dst = smem + slot * SLOT + r_dst[j];
			if (r_lse[j]) {
				const char* src = r_arr[j] + rowL * 4;
				__builtin_amdgcn_global_load_lds((gbl_void_p)(src + voff), (lds_void_p)dst, 4, 0, 0);
			} else {
				const char* src = r_arr[j] + rowI * r_ldb[j] + i_h * (2 * DH);
				__builtin_amdgcn_global_load_lds((gbl_void_p)(src + voff), (lds_void_p)dst, 16, 0, 0);
			}
		}
		if (i_u + 1 < QT2) {
			++i_u;
		} else if (i_pi + 1 < npairs) {
			++i_pi;
			i_u = 0;
			if (++i_h == n_head) {
				i_h = 0;
				++i_b;
			}
		}
	};
	// the K image of pair (b, h) into K buffer kb and its V image: NKV requests per wave (pieces past the last repeat it)
	auto issue_kv = [&](int b, int h, int kb) {
		const char* kbase = pk + ((int64_t)b * Skv * ldk + h * DH) * 2;
		const char* vbase = pv + ((int64_t)b * Skv * ldv + h * DH) * 2;
		int ln = lane;
		asm volatile("" : "+v"(ln));
#pragma unroll
		for (int t = 0; t < KVPW; ++t) {
			const int pc = min(wave + 8 * t, 4 * KT2 - 1);
			const int row = 8 * pc + (ln >> 3);
			const int rs = min(row, Skv - 1);
			const int inrow = ((ln & 7) ^ (row & 7)) << 4;
			__builtin_amdgcn_global_load_lds((gbl_void_p)(kbase + (uint32_t)(rs * 2 * ldk + inrow)), (lds_void_p)(smem + KOFF + kb * KIMG + pc * 1024), 16, 0, 0);
		}
#pragma unroll
		for (int t = 0; t < KVPW; ++t) {
			const int pc = min(wave + 8 * t, 4 * KT2 - 1);
			const int row = 8 * pc + (ln >> 3);
			const int rs = min(row, Skv - 1);
			const int inrow = ((ln & 7) ^ (row & 7)) << 4;
			__builtin_amdgcn_global_load_lds((gbl_void_p)(vbase + (uint32_t)(rs * 2 * ldv + inrow)), (lds_void_p)(smem + VOFF + pc * 1024), 16, 0, 0);
		}
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
	// region 2: the tiles of dQ^T (16 head columns x 16 queries; eight per step) that this wave forms: none, one or two.  The host
	// deals them to the waves with the least work in region 1 (13 key tiles on 8 waves: five waves own two, and the SIMD that
	// hosts two of those five gets no dQ tile)
	int ndq = 0, dqt[2] = {0, 0};
#pragma unroll
	for (int t = 0; t < 8; ++t)
		if ((int)((a.dq_owner >> (3 * t)) & 7u) == wave) {
			if (ndq < 2) dqt[ndq] = t;
			++ndq;
		}
	ndq = min(ndq, 2);
	asm volatile("" : "+s"(ndq));
	// dS block of a key tile: row i16 of the tile, the lane's four queries 16 hf + 4 g .. + 3 are 8 bytes at logical chunk
	// 4 parity + 2 hf + (g >> 1), half (g & 1)
	uint32_t dsw[2];
#pragma unroll
	for (int hf = 0; hf < 2; ++hf) dsw[hf] = (uint32_t)(i16 * 128 + (((2 * hf + (g >> 1)) ^ (i16 & 7)) << 4) + ((g & 1) << 3));

	// the row fragments of a query step (Q, dO, O: rows i16 of the two 16-query blocks; lse of the lane's queries 4 g + r):
	// requested one step ahead, right behind the barrier that publishes the slot
	bf16x8 fq[2][2], fg[2][2], fo[2][2];
	f32x4 l4[2];
	auto read_rows = [&](uint32_t sb) {
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
	};
	// ... and what region 1 needs of them per query besides the fragments themselves, formed at the end of region 2 (the reads
	// have long landed): delta[q] = sum_d dO[q][d] O[q][d], the diagonal of dO O^T (rows 4 g + r on this lane, column i16: the
	// diagonal element of row 4 g + r sits in lane 16 g + 4 g + r, register r; the cross-lane move is retired by the compiler
	// with lgkmcnt(0), made while no read issued here is in flight), and lse in log2 units
	float nd4[2][4], nlg[2][4];
	auto row_stats = [&]() {
		asm volatile("s_waitcnt lgkmcnt(0)"
					 : "+v"(fg[0][0]), "+v"(fg[0][1]), "+v"(fg[1][0]), "+v"(fg[1][1]), "+v"(fo[0][0]), "+v"(fo[0][1]), "+v"(fo[1][0]), "+v"(fo[1][1]), "+v"(l4[0]), "+v"(l4[1])
					 :
					 : "memory");
		f32x4 dd[2];
#pragma unroll
		for (int hf = 0; hf < 2; ++hf) {
			dd[hf] = mfma16(fg[hf][0], fo[hf][0], (f32x4){0.f, 0.f, 0.f, 0.f});
			dd[hf] = mfma16(fg[hf][1], fo[hf][1], dd[hf]);
		}
		const int dsrc = 20 * g;
#pragma unroll
		for (int hf = 0; hf < 2; ++hf)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				nd4[hf][r] = __shfl(dd[hf][r], dsrc + r, 64);
				nlg[hf][r] = l4[hf][r] * LOG2E;
			}
		asm volatile("" : "+v"(nd4[0][0]), "+v"(nd4[0][1]), "+v"(nd4[0][2]), "+v"(nd4[0][3]), "+v"(nd4[1][0]), "+v"(nd4[1][1]), "+v"(nd4[1][2]), "+v"(nd4[1][3])::"memory");
	};

	// ---- prologue -------------------------------------------------------------------------------------------------------
	// the dS image starts at zero: rows of key tiles that no wave owns (up to the padded key count) are read by region 2
	for (int i = tid; i < KIMG / 16; i += 512) reinterpret_cast<uint4*>(smem + DSOFF)[i] = make_uint4(0, 0, 0, 0);
	__syncthreads();
	int cb = i_b, ch = i_h;  // batch and head of the current pair
	issue_kv(cb, ch, 0);
	issue_ring(0);
	issue_ring(1);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__builtin_amdgcn_s_barrier();
	if (own0) {
		read_rows(lds0);
		if (VITK_ATTN_DELTA_AHEAD) row_stats();
	}

	if (VITK_ATTN_PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);  // the second-dispatched half loses every arbitration otherwise
#ifdef VITK_ATTN_STAMPS
	uint32_t seg[12] = {0};
	uint64_t tprev = __builtin_readcyclecounter();
	const uint64_t tstart = tprev;
#endif
	int s = 0, slot = 0;
#pragma clang loop unroll(disable)
	for (int pi = 0; pi < npairs; ++pi) {
		const int bh = bh0 + pi;
		const int kb = pi & 1;
		const uint32_t kimg = lds0 + KOFF + kb * KIMG, vimg = lds0 + VOFF, dsimg = lds0 + DSOFF;
		const uint32_t hk = DROP ? attn_head_key(a, bh) : 0u;
		// batch and head of the next pair (whose K / V images this pair requests); the last pair repeats itself
		int nb = cb, nh = ch;
		if (pi + 1 < npairs && ++nh == n_head) {
			nh = 0;
			++nb;
		}
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
			STAMP(0);  // pair start (K / V fragments) or loop overhead
			// ================= region 1: key on the lane =================
			if (own0) {
				if (!VITK_ATTN_DELTA_AHEAD) row_stats();
				asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fg[0][0]), "+v"(fg[0][1]), "+v"(fg[1][0]), "+v"(fg[1][1]), "+v"(fq[0][0]), "+v"(fq[0][1]), "+v"(fq[1][0]), "+v"(fq[1][1])::"memory");
				float d4[2][4], lg[2][4];
#pragma unroll
				for (int hf = 0; hf < 2; ++hf)
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						d4[hf][r] = nd4[hf][r];
						lg[hf][r] = nlg[hf][r];
					}
				STAMP(1);  // row fragments, delta
				if (u == QT2 - 1) {  // queries past the last one: lse = +inf makes their probabilities exactly zero
#pragma unroll
					for (int hf = 0; hf < 2; ++hf)
#pragma unroll
						for (int r = 0; r < 4; ++r)
							if (32 * u + 16 * hf + 4 * g + r >= Sq) lg[hf][r] = INFINITY;
				}
				uint32_t rk[2][4];
				if constexpr (DROP) {
#pragma unroll
					for (int hf = 0; hf < 2; ++hf)
#pragma unroll
						for (int r = 0; r < 4; ++r) rk[hf][r] = attn_row_key(hk, 32 * u + 16 * hf + 4 * g + r);
				}
				// S, dP, then P and dS (without the scale factor: dK and dQ are multiplied by it once, at the end) of one key tile;
				// the dS block goes to the LDS image
				bf16x8 pf[2], dsf[2];
				f32x4 svv[2][2], dpvv[2][2];
				auto tile_scores = [&](int j) {
#pragma unroll
					for (int hf = 0; hf < 2; ++hf) {
						svv[j][hf] = mfma16(fq[hf][0], ck[j][0], (f32x4){0.f, 0.f, 0.f, 0.f});
						dpvv[j][hf] = mfma16(fg[hf][0], cv[j][0], (f32x4){0.f, 0.f, 0.f, 0.f});
					}
#pragma unroll
					for (int hf = 0; hf < 2; ++hf) {
						svv[j][hf] = mfma16(fq[hf][1], ck[j][1], svv[j][hf]);
						dpvv[j][hf] = mfma16(fg[hf][1], cv[j][1], dpvv[j][hf]);
					}
				};
				auto tile_softmax = [&](int j) {
					if (!VITK_ATTN_TILES_TOGETHER) tile_scores(j);
					const f32x4(&sv)[2] = svv[j];
					const f32x4(&dpv)[2] = dpvv[j];
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
							ds[hf][r] = pr * ((DROP ? dpv[hf][r] * keep : dpv[hf][r]) - d4[hf][r]);
						}
					pf[j] = pack8(p[0], p[1]);
					dsf[j] = pack8(ds[0], ds[1]);
					// keys past the last one (their rows repeat it) contribute nothing to dQ: only the wave that owns the last tile runs
					// this (a branch on a wave-uniform value made opaque; as a select on every tile it was a quarter of the
					// arithmetic above)
					int ragged = __builtin_amdgcn_readfirstlane(16 * tile + 16 > Skv ? 1 : 0);
					asm volatile("" : "+s"(ragged));
					if (ragged) {
						union {
							bf16x8 v;
							uint32_t w[4];
						} zp, zd;
						zp.v = pf[j];
						zd.v = dsf[j];
						const bool past = key >= Skv;
#pragma unroll
						for (int e = 0; e < 4; ++e) {
							zp.w[e] = past ? 0u : zp.w[e];
							zd.w[e] = past ? 0u : zd.w[e];
						}
						pf[j] = zp.v;
						dsf[j] = zd.v;
					}
					union {
						bf16x8 v;
						s16x4 h[2];
					} w;
					w.v = dsf[j];
					const uint32_t rowb = dsimg + (uint32_t)tile * 2048u;
					wr64(rowb + (dsw[0] ^ (uint32_t)(par << 6)), w.h[0]);
					wr64(rowb + (dsw[1] ^ (uint32_t)(par << 6)), w.h[1]);
				};
				if (VITK_ATTN_TILES_TOGETHER) {
					tile_scores(0);
					if (own1) tile_scores(1);
				}
				tile_softmax(0);
				if (!VITK_ATTN_TILES_TOGETHER) __builtin_amdgcn_sched_barrier(0);
				STAMP(2);
				// dV^T += dO^T P, dK^T += Q^T dS: the transposed fragments in two halves of the head size, the first requested in
				// front of the second tile's arithmetic
				s16x4 gt[2][2][2], qt[2][2][2];
				auto tr_half = [&](int half) {
#pragma unroll
					for (int d2 = 0; d2 < 2; ++d2) {
						rdtr(qt[half][d2][0], sb + trb[2 * half + d2], 0);
						rdtr(qt[half][d2][1], sb + trb[2 * half + d2], 2048);
						rdtr(gt[half][d2][0], sb + trb[2 * half + d2], 4096);
						rdtr(gt[half][d2][1], sb + trb[2 * half + d2], 4096 + 2048);
					}
				};
				tr_half(0);
				if (own1) tile_softmax(1);
				__builtin_amdgcn_sched_barrier(0);
				STAMP(3);
				tr_half(1);
#pragma unroll
				for (int half = 0; half < 2; ++half) {
					// (the first half's eight reads, and the dS writes behind them, have returned once at most the second half's
					// eight are outstanding)
					if (half == 0)
						asm volatile("s_waitcnt lgkmcnt(8)"
									 : "+v"(gt[0][0][0]), "+v"(gt[0][0][1]), "+v"(gt[0][1][0]), "+v"(gt[0][1][1]), "+v"(qt[0][0][0]), "+v"(qt[0][0][1]), "+v"(qt[0][1][0]), "+v"(qt[0][1][1])
									 :
									 : "memory");
					else
						asm volatile("s_waitcnt lgkmcnt(0)"
									 : "+v"(gt[1][0][0]), "+v"(gt[1][0][1]), "+v"(gt[1][1][0]), "+v"(gt[1][1][1]), "+v"(qt[1][0][0]), "+v"(qt[1][0][1]), "+v"(qt[1][1][0]), "+v"(qt[1][1][1])
									 :
									 : "memory");
#pragma unroll
					for (int d2 = 0; d2 < 2; ++d2) {
						const int dt = 2 * half + d2;
						const bf16x8 ag = join8(gt[half][d2][0], gt[half][d2][1]), aq = join8(qt[half][d2][0], qt[half][d2][1]);
						dv[0][dt] = mfma16(ag, pf[0], dv[0][dt]);
						dk[0][dt] = mfma16(aq, dsf[0], dk[0][dt]);
						if (own1) {
							dv[1][dt] = mfma16(ag, pf[1], dv[1][dt]);
							dk[1][dt] = mfma16(aq, dsf[1], dk[1][dt]);
						}
					}
				}
				STAMP(4);  // dV / dK products
				if (u == QT2 - 1) {
					// This pair's dK, dV: 4 stores of 16 bytes per owned tile, each issued whether or not the lane's key exists.
					// A lane holds head columns 16 dt + 4 g .. + 3 of its key for dt = 0 .. 3; v_permlane16_swap_b32 exchanges
					// the odd 16-lane rows of one register with the even rows of another, after which a lane of an even row holds
					// columns 4 g .. 4 g + 7 of tile dt, a lane of an odd row columns 4 (g - 1) .. + 7 of tile dt + 1: 8
					// consecutive columns, 64 contiguous bytes per key and wave-instruction (16 stores of 8 bytes per tile, 32
					// contiguous bytes each, made the end of every pair a queue of stores: 3300 - 6000 cycles).
					int ln = lane;
					asm volatile("" : "+v"(ln));
					const int g2 = ln >> 4;
					const int col0 = (g2 & 1) ? 16 + 4 * (g2 - 1) : 4 * g2;
#pragma unroll
					for (int j = 0; j < 2; ++j) {
						if (j == 1 && !own1) continue;
						const int key = 16 * (wave + 8 * j) + (ln & 15);
						const bool ok = key < Skv;
						char* dkp = pdk + (((int64_t)cb * Skv + key) * lddk + ch * DH + col0) * 2;
						char* dvp = pdv + (((int64_t)cb * Skv + key) * lddv + ch * DH + col0) * 2;
#pragma unroll
						for (int d2 = 0; d2 < 2; ++d2) {
#pragma unroll
							for (int m = 0; m < 2; ++m) {
								const f32x4 e0 = m == 0 ? dk[j][2 * d2] * scale : dv[j][2 * d2], e1 = m == 0 ? dk[j][2 * d2 + 1] * scale : dv[j][2 * d2 + 1];
								union {
									bf16x4 v;
									uint32_t w[2];
								} ea, eb;
#pragma unroll
								for (int r = 0; r < 4; ++r) {
									ea.v[r] = (bf16_t)e0[r];
									eb.v[r] = (bf16_t)e1[r];
								}
								const auto s0 = __builtin_amdgcn_permlane16_swap(ea.w[0], eb.w[0], false, false);
								const auto s1 = __builtin_amdgcn_permlane16_swap(ea.w[1], eb.w[1], false, false);
								const u32x4 ov = {s0[0], s1[0], s0[1], s1[1]};
								char* dst = ok ? (m == 0 ? dkp : dvp) + 64 * d2 : dump;
								*(gbl_u32x4_p)dst = ov;  // (global address space spelled out: the opaque pointer copies would give flat stores)
							}
						}
					}
				}
			}
			// this wave's dS block is in LDS and its requests for the next step have landed; everything younger stays in flight:
			// the dQ stores of the previous step, the next pair's K / V requests if the previous step issued them (unless this
			// step ends the pair: then they must have landed too), and the dK / dV stores above
			STAMP(5);  // dK / dV stores
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			vm_wait(ndq + ((u == 1 && QT2 > 2) ? NKV : 0) + (u == QT2 - 1 ? nst : 0));
			STAMP(6);  // waiting for this wave's requests of the next step
			__builtin_amdgcn_s_barrier();
			STAMP(7);  // waiting for the other waves
			// ================= region 2: query on the lane =================
			const int nslot = slot == RING - 1 ? 0 : slot + 1;
			if (own0) read_rows(lds0 + nslot * SLOT);  // the next step's row fragments (published by the barrier above)
			issue_ring(slot == 0 ? RING - 1 : slot - 1);  // two steps ahead, into the slot of step s - 1
			if (u == 0) issue_kv(nb, nh, kb ^ 1);
			asm volatile("" ::: "memory");
			STAMP(8);  // request issue
#pragma unroll
			for (int it = 0; it < 2; ++it) {
				if (it >= ndq) break;
				const int qtw = dqt[it] >> 2, dtw = dqt[it] & 3;
				const uint32_t ab = kimg + tr_off(2 * dtw), bb = dsimg + tr_off(4 * par + 2 * qtw);
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
				char* dqp = pdq + (((int64_t)cb * Sq + q) * lddq + ch * DH + 16 * dtw + 4 * (ln >> 4)) * 2;
				bf16x4 ov;
#pragma unroll
				for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(dq[r] * scale);
				*(gbl_bf16x4_p)(q < Sq ? dqp : dump) = ov;
			}
			if (VITK_ATTN_DELTA_AHEAD && own0) row_stats();
			asm volatile("" ::: "memory");
			STAMP(9);  // dQ tiles, the next step's delta
			slot = nslot;
		}
		cb = nb;
		ch = nh;
	}
	asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the trailing (repeated) requests must not outlive the workgroup's LDS
#ifdef VITK_ATTN_STAMPS
	if (blockIdx.x < 64 && (tid & 63) == 0) {
		uint32_t* o = reinterpret_cast<uint32_t*>(a.bwd_scratch) + 4096 + (blockIdx.x * 8 + wave) * 16;
		for (int i = 0; i < 12; ++i) o[i] = seg[i];
		o[12] = (uint32_t)(__builtin_readcyclecounter() - tstart);
		o[13] = (uint32_t)(npairs * QT2);
	}
#endif
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
	{
		// the eight dQ tiles of a step go, one at a time, to the wave whose SIMD (waves w and w + 4 share one) and which itself
		// has the fewest matrix instructions per step so far; at most two per wave
		const int nk = (a.Skv + 15) / 16;
		int own[8], ntile[8] = {0};
		for (int w = 0; w < 8; ++w) {
			const int t = (w < nk ? 1 : 0) + (w + 8 < nk ? 1 : 0);
			own[w] = t > 0 ? 4 + 16 * t : 0;
		}
		a.dq_owner = 0;
		static int uniform = -1;
		if (uniform < 0) uniform = getenv("VITK_ATTN_DQ_UNIFORM") != nullptr ? 1 : 0;
		for (int t = 0; t < 8 && uniform; ++t) a.dq_owner |= (uint32_t)t << (3 * t);
		for (int t = 0; t < 8 && !uniform; ++t) {
			int best = -1, best_cost = 0;
			for (int w = 0; w < 8; ++w) {
				if (ntile[w] >= 2) continue;
				const int cost = 2 * (own[w & 3] + own[(w & 3) + 4]) + own[w];
				if (best < 0 || cost < best_cost) {
					best = w;
					best_cost = cost;
				}
			}
			a.dq_owner |= (uint32_t)best << (3 * t);
			own[best] += 7;
			++ntile[best];
		}
	}
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
