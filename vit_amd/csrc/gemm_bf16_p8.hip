// bf16 MFMA GEMM, K-major x K-major operands, bf16 output (forward and input-gradient GEMMs of nn.Linear,
// layers/transformers.py:46-54 and layers/attention.py:28,47 of the reference): persistent 256 x 256 x 64 tiles, one
// 512-thread workgroup per CU, built around a DEEP operand pipeline instead of one "wait for everything + barrier" per
// K step.
//
// Wave w = (wr, wc) = (w >> 2, w & 3) owns rows wr * 128 + [0, 128) and columns wc * 64 + [0, 64) of the tile: 8 x 4
// tiles of v_mfma_f32_16x16x32_bf16, 128 accumulator registers, cut into four QUADRANTS of 64 rows x 32 columns
// (row half h, column half h').  An operand K step (256 rows x 64 k) is staged as two HALF-TILES of 16 KiB: A_h holds
// row half h of BOTH wave rows (LDS row wr * 64 + i <-> tile row wr * 128 + h * 64 + i), B_h' column half h' of all four
// wave columns (LDS row wc * 32 + j <-> tile column wc * 64 + h' * 32 + j), so that one quadrant of every wave needs exactly
// one A and one B half-tile.  LDS holds two K steps (2 x 4 half-tiles, 128 KiB) + 8 x 4 KiB of epilogue images.
//
// A K step runs as two phases of 32 MFMAs per wave:
//     phase A: read B_0, B_1, A_0   row half 0 (quadrants (0,0), (0,1))     DMA: A_1 of K step u+1
//     phase B: read A_1             row half 1 (quadrants (1,0), (1,1))     DMA: A_0, B_0, B_1 of K step u+2
// Each phase refills (LDS-DMA, global_load_lds_dwordx4, two requests per lane and half-tile) the half-tiles whose last
// fragment read was retired in the phase before, for the K step that will use that buffer next.  The only waits on that
// stream are counted: s_waitcnt vmcnt(6) in phase A (all but the six newest requests -- A_0, B_0, B_1 of the next step --
// have landed, so this step's A_1 is in LDS) and vmcnt(2) in phase B (all but A_1 of the next step have landed, so the next
// step's A_0, B_0, B_1 are in LDS).  The memory pipeline never drains, and operand requests run 1.75 K steps ahead of
// their use, across tile boundaries too (the first K step of a later tile adds the previous epilogue's operation count
// to both waits, since those loads and stores are newer still and stay in flight).  (Four phases of 16 MFMAs were measured
// first: no faster than one barrier per K step, every barrier pair costs about 64 cycles.)
//
// The two wave rows (waves 0-3 / 4-7, one wave of each per SIMD) run half a phase apart: each phase is
// {fragment reads + DMA issue, barrier, 32 MFMAs, barrier}, and the second wave row starts one barrier late, so on every
// SIMD one wave is in its MFMA part while its partner is in its load part.
//
// Ordering rules the schedule obeys (an LDS-DMA is ordered for a reader only by the issuer's counted vmcnt followed by a
// barrier the reader passes; an LDS read is retired only by the reading wave's lgkmcnt):
//   - every wave retires a phase's fragment reads (lgkmcnt(0)) BEFORE that phase's first barrier, so a half-tile may be
//     refilled by any wave from the next phase on;
//   - each counted vmcnt sits before its phase's FIRST barrier, so that both wave rows have passed a barrier behind every
//     wave's wait before the first read of the half-tiles it covers.
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "gemm_common.h"
#include "gemm_epilogue_wave.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
constexpr int HALF_BYTES = 128 * BK * 2;      // 16 KiB
constexpr int STEP_BYTES = 4 * HALF_BYTES;    // [A_0][A_1][B_0][B_1]
constexpr int EPI_OFF = 2 * STEP_BYTES;
constexpr int EW_BYTES = 16 * 64 * 4;
constexpr int LDS_BYTES = EPI_OFF + 8 * EW_BYTES;  // 163,840 B: the whole LDS of a CU

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

__device__ __forceinline__ int off_k(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(lds_void_p)(void*)p; }

// 16-byte LDS read issued as inline asm (NOT waited for: the caller retires it with an s_waitcnt that names `dst`).  Plain
// LDS loads would do: but behind an LDS-DMA still in flight the compiler puts s_waitcnt vmcnt(0) in front of them.
__device__ __forceinline__ void lds_read128(bf16x8& dst, uint32_t addr, int offset) {
	asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(offset));
}

struct P8Args {
	GemmDev g;
	int tiles_m, tiles_n;
	int vec8;  // epilogue operands / outputs may be accessed 16 bytes wide
	int rows_per_xcd;  // tile order: whole tile rows per XCD instead of groups of four rows
#ifdef VITK_P8_STAMPS
	long long* stamps;
#endif
};

#ifdef VITK_P8_STAMPS
long long* g_p8_stamps = nullptr;
#define P8STAMP(slot) \
	if (a.stamps != nullptr && tid == 0 && round < 16) a.stamps[((int64_t)blockIdx.x * 16 + round) * 4 + (slot)] = __builtin_readcyclecounter()
#define PH(n) \
	if (a.stamps != nullptr && kt == 3 && round == 0) ph[n] = __builtin_readcyclecounter()
#else
#define P8STAMP(slot)
#define PH(n)
#endif

template <int EPI, int ACT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_p8_kernel(P8Args a) {
	// memory operations an epilogue issues at least (per wave-lane: 16 output stores, + 16 operand loads with an aux operand)
	constexpr int N_EPI = (EPI == VITK_EPI_RESIDUAL || EPI == VITK_EPI_ACTGRAD) ? 32 : 16;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	gemm_dev_enter(a.g);
	const GemmDev& g = a.g;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wr = wave >> 2, wc = wave & 3;
	const int ntiles = a.tiles_m * a.tiles_n;
	const int nblk = gridDim.x;  // a multiple of 8
	const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
	constexpr int GM = 4;
	const int per_group = GM * a.tiles_n;

	// tile of this workgroup in a round: consecutive tiles (in grouped order) go to workgroups that share an XCD
	auto tile_of = [&](int round, int64_t& m0, int64_t& n0) -> bool {
		if (a.rows_per_xcd) {
			// whole tile rows to one XCD (tile row r belongs to XCD r % 8, its tiles to consecutive workgroups of that XCD): an A
			// panel is then fetched into ONE L2
			const int per = nblk >> 3;  // workgroups of an XCD
			const int rows_x = (a.tiles_m - xcd + 7) >> 3;  // tile rows of this XCD
			const int t = round * per + idx;
			if (t >= rows_x * a.tiles_n) return false;
			const int rr = t / a.tiles_n;
			m0 = (int64_t)(xcd + 8 * rr) * BM;
			n0 = (int64_t)(t - rr * a.tiles_n) * BN;
			return true;
		}
		const int base = round * nblk;
		const int n = min(nblk, ntiles - base);
		if (n <= 0) return false;
		const int q = n >> 3, r = n & 7;
		if (idx >= q + (xcd < r ? 1 : 0)) return false;
		const int t = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
		const int group = t / per_group;
		const int first_m = group * GM;
		const int gsz = min(a.tiles_m - first_m, GM);
		const int tm = first_m + (t % per_group) % gsz;
		const int tn = (t % per_group) / gsz;
		m0 = (int64_t)tm * BM;
		n0 = (int64_t)tn * BN;
		return true;
	};

	const int nt = (int)(g.K / BK);  // >= 2
	const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
	const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
	char* const img = smem + EPI_OFF + wave * EW_BYTES;
	const uint32_t lds0 = lds_addr(smem);

	// LDS-DMA source offset of a lane inside an 8-row piece (row lane >> 3, 16-byte chunk (lane & 7) ^ row)
	const int pr = lane >> 3, pch = (lane & 7) ^ (lane >> 3);
	const uint32_t a_voff = (uint32_t)(pr * (int)g.lda + 8 * pch) * 2u;
	const uint32_t b_voff = (uint32_t)(pr * (int)g.ldb + 8 * pch) * 2u;

	// One half-tile: 16 pieces of 8 rows, two per wave.  which: 0 = A_0, 1 = A_1, 2 = B_0, 3 = B_1.
	auto stage_half = [&](int buf, int which, int64_t tm0, int64_t tn0, int k0) {
		char* dst = smem + buf * STEP_BYTES + which * HALF_BYTES;
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const int p = 2 * wave + j;
			if (which < 2) {
				const int64_t row0 = tm0 + (p >> 3) * 128 + which * 64 + 8 * (p & 7);
				const char* src = reinterpret_cast<const char*>(A + row0 * g.lda + k0) + a_voff;
				__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(dst + p * 1024), 16, 0, 0);
			} else {
				// (N is a multiple of 64, not necessarily of 256: a piece past the last column re-fetches the last 8 columns, and the
				// waves that own those columns skip their epilogue)
				const int64_t col0 = min(tn0 + (p >> 2) * 64 + (which - 2) * 32 + 8 * (p & 3), g.N - 8);
				const char* src = reinterpret_cast<const char*>(B + col0 * g.ldb + k0) + b_voff;
				__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(dst + p * 1024), 16, 0, 0);
			}
		}
	};

	// fragment read addresses inside a half-tile: row (lane & 15) of a 16-row block, 16-byte chunk (lane >> 4) ^ (row & 7);
	// the second half of the K step is the same address with bit 6 flipped (chunk + 4), the next 16 rows are +2048
	const uint32_t a_lane = (uint32_t)off_k(wr * 64 + (lane & 15), lane >> 4);
	const uint32_t b_lane = (uint32_t)off_k(wc * 32 + (lane & 15), lane >> 4);

	int64_t m0 = 0, n0 = 0;
	bool have = tile_of(0, m0, n0);
	if (!have) return;  // (grids never exceed the tile count)
	int64_t nm0 = m0, nn0 = n0;  // the tile after this one; after the last tile its operands are fetched again (nobody reads them)
	bool nxt = tile_of(1, nm0, nn0);

	// K step `kt + d` of the current tile, or the matching step of the next tile
	auto stage_ahead = [&](int buf, int which, int kt, int d) {
		const int k = kt + d;
		const bool cur = k < nt;
		stage_half(buf, which, cur ? m0 : nm0, cur ? n0 : nn0, (cur ? k : k - nt) * BK);
	};

	// prologue: all of K step 0, and A_0, B_0, B_1 of K step 1 (its A_1 goes out in phase A of step 0)
	stage_ahead(0, 2, 0, 0);
	stage_ahead(0, 0, 0, 0);
	stage_ahead(0, 3, 0, 0);
	stage_ahead(0, 1, 0, 0);
	stage_ahead(1, 0, 0, 1);
	stage_ahead(1, 2, 0, 1);
	stage_ahead(1, 3, 0, 1);
	asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
	__builtin_amdgcn_s_barrier();

	int par = 0;  // LDS buffer of the current K step
	bool epi_pending = false;  // this wave's previous epilogue left its N_EPI memory operations in flight (wave-uniform)
	for (int round = 0; have; ++round) {
		P8STAMP(0);
		f32x4 acc[8][4];
#pragma unroll
		for (int i = 0; i < 8; ++i)
#pragma unroll
			for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
		if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs half a phase behind the first
#ifdef VITK_P8_STAMPS
		long long ph[5] = {0};
#endif
		for (int kt = 0; kt < nt; ++kt) {
			PH(0);
			const uint32_t sb = lds0 + par * STEP_BYTES;
			bf16x8 xa[4][2], xb[4][2];
			// ---- phase A: B_0, B_1, A_0 -> row half 0 (quadrants (0,0), (0,1)); DMA: A_1 of the next K step ----
#pragma unroll
			for (int j = 0; j < 4; ++j)
#pragma unroll
				for (int s = 0; s < 2; ++s) lds_read128(xb[j][s], (sb + (2 + (j >> 1)) * HALF_BYTES + b_lane) ^ (s << 6), 2048 * (j & 1));
#pragma unroll
			for (int mi = 0; mi < 4; ++mi)
#pragma unroll
				for (int s = 0; s < 2; ++s) lds_read128(xa[mi][s], (sb + a_lane) ^ (s << 6), 2048 * mi);
			// all but the six newest requests (A_0, B_0, B_1 of the next K step) have landed: this step's A_1 is in LDS.  In
			// the first K step of a tile the previous epilogue's loads and stores are newer still and stay in flight too.
			if (kt == 0 && epi_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + N_EPI) : "memory");
			else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
			stage_ahead(par ^ 1, 1, kt, 1);
			asm volatile("s_waitcnt lgkmcnt(0)"
						 : "+v"(xb[0][0]), "+v"(xb[0][1]), "+v"(xb[1][0]), "+v"(xb[1][1]), "+v"(xb[2][0]), "+v"(xb[2][1]), "+v"(xb[3][0]), "+v"(xb[3][1]), "+v"(xa[0][0]),
						   "+v"(xa[0][1]), "+v"(xa[1][0]), "+v"(xa[1][1]), "+v"(xa[2][0]), "+v"(xa[2][1]), "+v"(xa[3][0]), "+v"(xa[3][1])
						 :
						 : "memory");
			__builtin_amdgcn_s_barrier();
			PH(1);
			__builtin_amdgcn_sched_barrier(0);
			__builtin_amdgcn_s_setprio(1);
#pragma unroll
			for (int s = 0; s < 2; ++s)
#pragma unroll
				for (int mi = 0; mi < 4; ++mi)
#pragma unroll
					for (int j = 0; j < 4; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb[j][s], xa[mi][s], acc[mi][j], 0, 0, 0);
			__builtin_amdgcn_s_setprio(0);
			__builtin_amdgcn_sched_barrier(0);
			__builtin_amdgcn_s_barrier();
			PH(2);
			// ---- phase B: A_1 -> row half 1 (quadrants (1,0), (1,1)); DMA: A_0, B_0, B_1 of the K step after next ----
#pragma unroll
			for (int mi = 0; mi < 4; ++mi)
#pragma unroll
				for (int s = 0; s < 2; ++s) lds_read128(xa[mi][s], (sb + HALF_BYTES + a_lane) ^ (s << 6), 2048 * mi);
			// all but the two newest requests (A_1 of the next K step) have landed: the next K step's A_0, B_0, B_1 are in LDS
			if (kt == 0 && epi_pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + N_EPI) : "memory");
			else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
			stage_ahead(par, 0, kt, 2);
			stage_ahead(par, 2, kt, 2);
			stage_ahead(par, 3, kt, 2);
			asm volatile("s_waitcnt lgkmcnt(0)"
						 : "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[1][0]), "+v"(xa[1][1]), "+v"(xa[2][0]), "+v"(xa[2][1]), "+v"(xa[3][0]), "+v"(xa[3][1])
						 :
						 : "memory");
			__builtin_amdgcn_s_barrier();
			PH(3);
			__builtin_amdgcn_sched_barrier(0);
			__builtin_amdgcn_s_setprio(1);
#pragma unroll
			for (int s = 0; s < 2; ++s)
#pragma unroll
				for (int mi = 0; mi < 4; ++mi)
#pragma unroll
					for (int j = 0; j < 4; ++j) acc[4 + mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb[j][s], xa[mi][s], acc[4 + mi][j], 0, 0, 0);
			__builtin_amdgcn_s_setprio(0);
			__builtin_amdgcn_sched_barrier(0);
			__builtin_amdgcn_s_barrier();
			PH(4);
			par ^= 1;
		}
		// both wave rows run their epilogues side by side: the first waits here for the second's last MFMA part
		if (wr == 0) __builtin_amdgcn_s_barrier();
		P8STAMP(1);
		asm volatile("" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
		const int64_t m0w = m0 + wr * 128, n0w = n0 + wc * 64;
		// (the host only launches this kernel on whole row tiles and 64-column wave slices with 16-byte accessible operands; a
		// wave whose slice lies past the last column has nothing to store)
		epi_pending = n0w < g.N;
		if (epi_pending) {
			// (an opaque copy of the lane index: lane-dependent epilogue addresses formed from the original are hoisted above
			// the main loop, spilled there, and their reload here waits for every operand request in flight)
			int lane_e = lane;
			asm volatile("" : "+v"(lane_e));
			EpiRegs er;
			epilogue_prefetch<EPI>(g, er, true, n0w, lane_e);
			epilogue_wave<EPI, ACT, false>(g, acc, er, true, img, m0w, n0w, lane_e);
		}
		P8STAMP(2);
#ifdef VITK_P8_STAMPS
		if (a.stamps != nullptr && round == 0 && lane == 0 && blockIdx.x == 8)
			for (int i = 0; i < 5; ++i) a.stamps[1024 * 16 * 4 + wave * 16 + i] = ph[i];
#endif
		m0 = nm0;
		n0 = nn0;
		have = nxt;
		nxt = tile_of(round + 2, nm0, nn0);  // (leaves nm0, nn0 alone when there is none)
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last (unused) DMA must not outlive the workgroup's LDS
}

// ---- weight gradients: dW[m, n] = sum over tokens k of dY[k, m] X[k, n] ----------------------------------------------------
// Same pipeline with MN-major operands (A(m, k) = A[k * lda + m]): a half-tile is [64 k][128 columns] in 256-byte rows, staged
// in pieces of 4 rows, fragments come out of it through the transposing read ds_read_b64_tr_b16 (two per fragment).  Bank
// swizzle: the 32-byte unit index of a row is XORed with fk(k) = (k & 3) | ((k >> 3) & 1) << 2, which is distinct for the 8
// rows one half-wave's transposing read touches.  One (tile, K slice) per workgroup, the float32 partial tile leaves in
// accumulator order (one contiguous KiB per wave-instruction).
__device__ __forceinline__ int fk(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

__device__ __forceinline__ void lds_read_tr(s16x4& dst, uint32_t addr, int offset) {
	asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(offset));
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

__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_p8w_kernel(WgradArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const GemmDev& g = a.g;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wr = wave >> 2, wc = wave & 3;
	// (tile, K slice) of this workgroup, K slice outermost: workgroups that share an XCD get neighbouring tiles of one slice
	const int ntile = a.tiles_m * a.tiles_n;
	const int nwg = ntile * a.splits;
	int wg = blockIdx.x;
	{
		const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
		wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
	}
	const int split = wg / ntile;
	wg -= split * ntile;
	constexpr int GM = 4;
	const int per_group = GM * a.tiles_n;
	const int group = wg / per_group;
	const int first_m = group * GM;
	const int gsz = min(a.tiles_m - first_m, GM);
	const int tm = first_m + (wg % per_group) % gsz;
	const int tn = (wg % per_group) / gsz;
	const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
	const int k_begin = split * a.k_per_split;
	const int k_end = min((int)g.K, k_begin + a.k_per_split);
	const int nt = (k_end - k_begin) / BK;  // >= 1

	const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
	const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
	const uint32_t lds0 = lds_addr(smem);

	// LDS-DMA: a piece is 4 k-rows x 256 bytes; lane l lands at row l >> 4, physical 16-byte chunk l & 15 and therefore
	// fetches the logical chunk whose unit is (physical unit) ^ fk(row).  Rows of a piece: k = 4 * piece + (l >> 4) with piece =
	// 2 * wave + j, so fk(k) = ((l >> 4) & 3) | (wave & 1) << 2 for both pieces of a wave: the lane offsets are loop invariant.
	const int prow = lane >> 4, pchunk = lane & 15;
	const int lchunk = ((((pchunk >> 1) ^ (prow | ((wave & 1) << 2))) << 1) | (pchunk & 1));  // logical chunk: 8 columns of the half-tile
	// half-tile column 8 * lchunk -> tile column: A halves interleave 64-column blocks, B halves 32-column blocks
	const int a_col = (lchunk >> 3) * 128 + 8 * (lchunk & 7);   // + h * 64
	const int b_col = (lchunk >> 2) * 64 + 8 * (lchunk & 3);    // + h' * 32
	const uint32_t a_voff = (uint32_t)(prow * (int)g.lda + a_col) * 2u;
	const uint32_t b_voff = (uint32_t)(prow * (int)g.ldb + b_col) * 2u;
	auto stage_half = [&](int buf, int which, int kstep) {
		char* dst = smem + buf * STEP_BYTES + which * HALF_BYTES;
		const int64_t k0 = k_begin + (int64_t)min(kstep, nt - 1) * BK;  // past the slice: fetch its last step again (nobody reads it)
#pragma unroll
		for (int j = 0; j < 2; ++j) {
			const int p = 2 * wave + j;
			const char* src;
			if (which < 2) src = reinterpret_cast<const char*>(A + (k0 + 4 * p) * g.lda + m0 + which * 64) + a_voff;
			else src = reinterpret_cast<const char*>(B + (k0 + 4 * p) * g.ldb + n0 + (which - 2) * 32) + b_voff;
			__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(dst + p * 1024), 16, 0, 0);
		}
	};

	// transposing fragment reads: lane (g4, q, p) = (lane >> 4, (lane >> 2) & 3, lane & 3) addresses k-row 8 g4 + q (+ 32 s, + 4 for
	// the second read), columns 4 p .. 4 p + 3 of the fragment's 16; fragment f of a wave starts at 16-byte chunk 2 f + (p >> 1)
	// of the wave's block, i.e. unit (block unit) ^ f: one address per operand, XOR f << 5 per fragment
	const int g4 = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
	const int frow = 8 * g4 + fq;
	auto frag_addr = [&](int unit0) { return (uint32_t)(frow * 256 + ((((unit0 ^ fk(frow)) << 1) | (fp >> 1)) << 4) + ((fp & 1) << 3)); };
	const uint32_t a_lane = frag_addr(wr * 4), b_lane = frag_addr(wc * 2);

	f32x4 acc[8][4];
#pragma unroll
	for (int i = 0; i < 8; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

	// prologue: all of K step 0, and A_0, B_0, B_1 of K step 1 (its A_1 goes out in phase A of step 0)
	stage_half(0, 2, 0);
	stage_half(0, 0, 0);
	stage_half(0, 3, 0);
	stage_half(0, 1, 0);
	stage_half(1, 0, 1);
	stage_half(1, 2, 1);
	stage_half(1, 3, 1);
	asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
	__builtin_amdgcn_s_barrier();

	// column sums of A (the bias gradient): the tiles_n workgroups that share an A tile take turns per K step; thread (c, r) =
	// (tid & 255, tid >> 8) adds up k-rows 32 r .. 32 r + 31 of tile column c, which lives in half-tile (c >> 6) & 1 =
	// wave & 1: even waves read A_0 in phase A, odd waves A_1 in phase B
	float csum = 0.f;
	const bool do_colsum = a.cs_partials != nullptr;
	const int cs_c = tid & 255;
	const int cs_hc = (cs_c >> 7) * 64 + (cs_c & 63);  // column inside the half-tile
	auto colsum_step = [&](uint32_t half_base) {
#pragma unroll
		for (int b4 = 0; b4 < 4; ++b4) {
			uint32_t v[8];
#pragma unroll
			for (int e = 0; e < 8; ++e) {
				const int kr = (tid >> 8) * 32 + 8 * b4 + e;
				const uint32_t ad = half_base + kr * 256 + (((((cs_hc >> 4) ^ fk(kr)) << 1) | ((cs_hc >> 3) & 1)) << 4) + (cs_hc & 7) * 2;
				asm volatile("ds_read_u16 %0, %1" : "=v"(v[e]) : "v"(ad));
			}
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
#pragma unroll
			for (int e = 0; e < 8; ++e) csum += __builtin_bit_cast(float, v[e] << 16);
		}
	};

	if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave row runs half a phase behind the first
	int par = 0;
	for (int kt = 0; kt < nt; ++kt) {
		const uint32_t sb = lds0 + par * STEP_BYTES;
		const bool cs_now = do_colsum && (k_begin / BK + kt) % a.tiles_n == tn;
		s16x4 xa[4][2][2], xb[4][2][2];  // [fragment][k half][read]
		// ---- phase A: B_0, B_1, A_0 -> row half 0; DMA: A_1 of the next K step ----
		if (cs_now && (wave & 1) == 0) colsum_step(sb);
#pragma unroll
		for (int j = 0; j < 4; ++j)
#pragma unroll
			for (int s = 0; s < 2; ++s) {
				lds_read_tr(xb[j][s][0], (sb + (2 + (j >> 1)) * HALF_BYTES + b_lane) ^ ((j & 1) << 5), 8192 * s);
				lds_read_tr(xb[j][s][1], (sb + (2 + (j >> 1)) * HALF_BYTES + b_lane) ^ ((j & 1) << 5), 8192 * s + 1024);
			}
#pragma unroll
		for (int mi = 0; mi < 4; ++mi)
#pragma unroll
			for (int s = 0; s < 2; ++s) {
				lds_read_tr(xa[mi][s][0], (sb + a_lane) ^ (mi << 5), 8192 * s);
				lds_read_tr(xa[mi][s][1], (sb + a_lane) ^ (mi << 5), 8192 * s + 1024);
			}
		asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
		stage_half(par ^ 1, 1, kt + 1);
		asm volatile("s_waitcnt lgkmcnt(0)"
					 : "+v"(xb[0][0][0]), "+v"(xb[0][0][1]), "+v"(xb[0][1][0]), "+v"(xb[0][1][1]), "+v"(xb[1][0][0]), "+v"(xb[1][0][1]), "+v"(xb[1][1][0]), "+v"(xb[1][1][1]),
					   "+v"(xb[2][0][0]), "+v"(xb[2][0][1]), "+v"(xb[2][1][0]), "+v"(xb[2][1][1]), "+v"(xb[3][0][0]), "+v"(xb[3][0][1]), "+v"(xb[3][1][0]), "+v"(xb[3][1][1])
					 :
					 : "memory");
		asm volatile(""
					 : "+v"(xa[0][0][0]), "+v"(xa[0][0][1]), "+v"(xa[0][1][0]), "+v"(xa[0][1][1]), "+v"(xa[1][0][0]), "+v"(xa[1][0][1]), "+v"(xa[1][1][0]), "+v"(xa[1][1][1]),
					   "+v"(xa[2][0][0]), "+v"(xa[2][0][1]), "+v"(xa[2][1][0]), "+v"(xa[2][1][1]), "+v"(xa[3][0][0]), "+v"(xa[3][0][1]), "+v"(xa[3][1][0]), "+v"(xa[3][1][1])
					 :
					 : "memory");
		__builtin_amdgcn_s_barrier();
		__builtin_amdgcn_sched_barrier(0);
		__builtin_amdgcn_s_setprio(1);
#pragma unroll
		for (int s = 0; s < 2; ++s)
#pragma unroll
			for (int mi = 0; mi < 4; ++mi)
#pragma unroll
				for (int j = 0; j < 4; ++j)
					acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(xb[j][s][0], xb[j][s][1]), join8(xa[mi][s][0], xa[mi][s][1]), acc[mi][j], 0, 0, 0);
		__builtin_amdgcn_s_setprio(0);
		__builtin_amdgcn_sched_barrier(0);
		__builtin_amdgcn_s_barrier();
		// ---- phase B: A_1 -> row half 1; DMA: A_0, B_0, B_1 of the K step after next ----
		if (cs_now && (wave & 1) == 1) colsum_step(sb + HALF_BYTES);
#pragma unroll
		for (int mi = 0; mi < 4; ++mi)
#pragma unroll
			for (int s = 0; s < 2; ++s) {
				lds_read_tr(xa[mi][s][0], (sb + HALF_BYTES + a_lane) ^ (mi << 5), 8192 * s);
				lds_read_tr(xa[mi][s][1], (sb + HALF_BYTES + a_lane) ^ (mi << 5), 8192 * s + 1024);
			}
		asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
		stage_half(par, 0, kt + 2);
		stage_half(par, 2, kt + 2);
		stage_half(par, 3, kt + 2);
		asm volatile("s_waitcnt lgkmcnt(0)"
					 : "+v"(xa[0][0][0]), "+v"(xa[0][0][1]), "+v"(xa[0][1][0]), "+v"(xa[0][1][1]), "+v"(xa[1][0][0]), "+v"(xa[1][0][1]), "+v"(xa[1][1][0]), "+v"(xa[1][1][1]),
					   "+v"(xa[2][0][0]), "+v"(xa[2][0][1]), "+v"(xa[2][1][0]), "+v"(xa[2][1][1]), "+v"(xa[3][0][0]), "+v"(xa[3][0][1]), "+v"(xa[3][1][0]), "+v"(xa[3][1][1])
					 :
					 : "memory");
		__builtin_amdgcn_s_barrier();
		__builtin_amdgcn_sched_barrier(0);
		__builtin_amdgcn_s_setprio(1);
#pragma unroll
		for (int s = 0; s < 2; ++s)
#pragma unroll
			for (int mi = 0; mi < 4; ++mi)
#pragma unroll
				for (int j = 0; j < 4; ++j)
					acc[4 + mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(xb[j][s][0], xb[j][s][1]), join8(xa[mi][s][0], xa[mi][s][1]), acc[4 + mi][j], 0, 0, 0);
		__builtin_amdgcn_s_setprio(0);
		__builtin_amdgcn_sched_barrier(0);
		__builtin_amdgcn_s_barrier();
		par ^= 1;
	}
	if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with the second wave row's last barrier
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing (unused) DMA must not outlive the workgroup's LDS
	if (do_colsum) a.cs_partials[(((int64_t)split * a.tiles_n + tn) * 2 + (tid >> 8)) * g.M + m0 + cs_c] = csum;
	// K-split partial tile straight from the accumulators in their own order; split_reduce_kernel adds the slices up
	float* dst = a.partials + ((int64_t)split * ntile + (int64_t)tm * a.tiles_n + tn) * (BM * BN) + (int64_t)wave * (32 * 256) + lane * 4;
#pragma unroll
	for (int i = 0; i < 8; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(dst + (i * 4 + j) * 256) = acc[i][j];
}

int p8_blocks() {
	static int n = 0;
	if (n == 0) {
		int dev = 0, cus = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
		n = cus & ~7;
	}
	return n;
}

// 1: a persistent launch uses the fewest workgroups that keep its round count (see launch_p8).  Measured on the ViT-B/16
// step: 0.2 ms slower when nothing else runs, 4 % faster when another kernel holds 16 CUs half of the time: off by default,
// switched on by the data-parallel trainer (vitk_gemm_leave_cus_free), VITK_P8_EVEN_ROUNDS=0/1 overrides both.
std::atomic<int> g_even_rounds{0};

bool p8_even_rounds() {
	static int env = -1;
	if (env < 0) {
		const char* e = getenv("VITK_P8_EVEN_ROUNDS");
		env = e == nullptr ? 2 : (e[0] != '0' ? 1 : 0);
	}
	return env == 2 ? g_even_rounds.load(std::memory_order_relaxed) != 0 : env == 1;
}

template <int EPI, int ACT>
int launch_p8(const P8Args& a, hipStream_t st) {
	static bool attr = false;
	if (!attr) {
		hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_p8_kernel<EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr = true;
	}
	const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
	int blocks = p8_blocks();
	if (tiles < blocks) {
		blocks = (int)((tiles + 7) & ~7ll);
	} else if (p8_even_rounds()) {
		// the fewest workgroups that still finish in the same number of rounds: 591 tiles take three rounds on 256 CUs and on 197,
		// and the CUs left alone run whatever else is queued (communication kernels during an overlapped all-reduce) without
		// delaying a workgroup of this launch
		const int64_t rounds = vitk_cdiv(tiles, (int64_t)blocks);
		blocks = (int)vitk_cdiv(tiles, rounds);
	}
	hipLaunchKernelGGL((gemm_bf16_p8_kernel<EPI, ACT>), dim3((unsigned)blocks), dim3(NTHREADS), LDS_BYTES, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

}  // namespace

#ifdef VITK_P8_STAMPS
void vitk_p8_set_stamps(long long* p) { g_p8_stamps = p; }
#endif

extern "C" int vitk_gemm_leave_cus_free(int on) { return g_even_rounds.exchange(on ? 1 : 0, std::memory_order_relaxed); }

bool vitk_gemm_bf16_p8_ok(const vitk_gemm_desc& d) {
	if (d.patch_grid > 0 || d.tokens_per_image > 0 || d.K % BK != 0 || d.K < 2 * BK) return false;
	if (d.a_layout != VITK_K_MAJOR || d.b_layout != VITK_K_MAJOR) return false;
	if (d.in_dtype != VITK_BF16 || d.out_dtype != VITK_BF16 || d.accumulate) return false;
	// whole row tiles; columns in whole 64-wide wave slices, at least half of the last tile column in use
	if (d.M % BM != 0 || d.N % 64 != 0 || 2 * d.N < vitk_cdiv(d.N, BN) * BN) return false;
	auto al16 = [](const void* p) { return p == nullptr || vitk_aligned(p, 16); };
	if (d.ldc % 8 != 0 || d.ld_aux % 8 != 0 || d.ld_out2 % 8 != 0 || !al16(d.C) || !al16(d.bias) || !al16(d.gamma) || !al16(d.aux) || !al16(d.out2)) return false;
	if (d.lda % 8 != 0 || d.ldb % 8 != 0 || !al16(d.A) || !al16(d.B)) return false;
	return true;
}

int vitk_gemm_bf16_p8(const vitk_gemm_desc& d, hipStream_t st) {
	vitk_last_gemm_kernel = "gemm_bf16_p8_kernel";
	P8Args a{};
	a.g = make_gemm_dev(d);
	a.tiles_m = (int)vitk_cdiv(d.M, BM);
	a.tiles_n = (int)vitk_cdiv(d.N, BN);
#ifdef VITK_P8_STAMPS
	a.stamps = g_p8_stamps;
#endif
	if (!vitk_gemm_bf16_p8_ok(d)) return VITK_ERR_UNSUPPORTED;
	a.vec8 = 1;
	{
		static int env = -1;
		if (env < 0) {
			const char* e = getenv("VITK_P8_ROWS_PER_XCD");
			env = (e != nullptr && e[0] == '1') ? 1 : 0;
		}
		a.rows_per_xcd = env && (int64_t)a.tiles_m * a.tiles_n >= p8_blocks() && !p8_even_rounds();
	}
	switch (d.epilogue) {
		case VITK_EPI_NONE: return launch_p8<VITK_EPI_NONE, -1>(a, st);
		case VITK_EPI_ACT: return d.act == VITK_ACT_GELU ? launch_p8<VITK_EPI_ACT, VITK_ACT_GELU>(a, st) : launch_p8<VITK_EPI_ACT, -1>(a, st);
		case VITK_EPI_RESIDUAL: return launch_p8<VITK_EPI_RESIDUAL, -1>(a, st);
		case VITK_EPI_ACTGRAD: return d.act == VITK_ACT_GELU ? launch_p8<VITK_EPI_ACTGRAD, VITK_ACT_GELU>(a, st) : launch_p8<VITK_EPI_ACTGRAD, -1>(a, st);
		default: return VITK_ERR_UNSUPPORTED;
	}
}

bool vitk_gemm_bf16_p8w_ok(const WgradArgs& a) {
	const GemmDev& g = a.g;
	if (a.partials == nullptr || a.splits < 1) return false;
	if (g.M % BM != 0 || g.N % BN != 0 || g.K % BK != 0 || a.k_per_split % BK != 0) return false;
	if (g.lda % 8 != 0 || g.ldb % 8 != 0 || !vitk_aligned(g.A, 16) || !vitk_aligned(g.B, 16)) return false;
	const int64_t last = g.K - (int64_t)(a.splits - 1) * a.k_per_split;  // the last K slice may be shorter
	if (a.k_per_split < BK || last < BK) return false;  // (a one-step slice works: requests past the slice re-fetch its last step)
	if ((int64_t)a.tiles_m * a.tiles_n * a.splits > (1ll << 30)) return false;
	return true;
}

int vitk_gemm_bf16_p8w(const WgradArgs& a, hipStream_t st) {
	if (!vitk_gemm_bf16_p8w_ok(a)) return VITK_ERR_UNSUPPORTED;
	vitk_last_gemm_kernel = "gemm_bf16_p8w_kernel";
	static bool attr = false;
	if (!attr) {
		hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_p8w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr = true;
	}
	hipLaunchKernelGGL(gemm_bf16_p8w_kernel, dim3((unsigned)(a.tiles_m * a.tiles_n * a.splits)), dim3(NTHREADS), LDS_BYTES, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}
