// bf16 GEMM on the CDNA4 matrix cores (v_mfma_f32_16x16x32_bf16), float32 accumulation, fused epilogue.
//
//   C[M,N] = epilogue( sum_k A(m,k) B(n,k) )
//
// 128x128x64 tile per 256-thread workgroup (2x2 waves, 64x64 per wave = 4x4 MFMA tiles), two LDS
// stages, one barrier per K-tile: the global loads of tile t+1 are issued into registers before the
// MFMAs of tile t and written to the other LDS stage after them.
//
// Operand tiles in LDS (16 KiB each):
//   K-major operand  (k contiguous in memory): image [128 rows][64 k] with 128-byte rows, the 16-byte
//     chunk index XOR-ed with (row & 7) so that the ds_read_b128 of a fragment (16 rows x one chunk
//     per lane group) is bank-conflict free.
//   MN-major operand (m or n contiguous; the dgrad weight and both wgrad operands): image
//     [64 k][128 cols] with 256-byte rows, read with the transposing ds_read_b64_tr_b16 so that no
//     transposed copy of activations or weights is ever materialised in HBM; 32-byte units are
//     XOR-ed with (k & 3) | ((k >> 3) & 1) << 2 to spread the 8 rows of a half-wave over the banks.
//
// The MFMA is issued as D = Wfrag x Xfrag so that a lane ends up with 4 consecutive n of one row m
// (8-byte bf16 / 16-byte f32 stores and epilogue operand loads).
#include <cstdlib>

#include "gemm_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB, both layouts
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int CT_LD = 132;                                   // floats per row of the epilogue image (128 + 4 pad)
constexpr int LDS_BYTES = 128 * CT_LD * 4;                   // 67,584 B >= the two operand stages (65,536 B)
constexpr int DEEP_STAGES = 4;                               // the deep-pipeline form: four stages, one workgroup per CU
constexpr int DEEP_LDS_BYTES = DEEP_STAGES * STAGE_BYTES;    // 131,072 B

__device__ __forceinline__ int swz_k(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ int swz_mn(int krow, int chunk) {
	const int unit = (chunk >> 1) ^ ((krow & 3) | (((krow >> 3) & 1) << 2));
	return krow * 256 + (((unit << 1) | (chunk & 1)) << 4);
}

struct Frag {
	bf16x8 v;
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// fragment read: 16 rows (or cols) starting at `base`, k-step s (32 wide) of the staged 64-wide K tile
template <int LAYOUT>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int base, int s, int lane) {
	if (LAYOUT == VITK_K_MAJOR) {
		const int row = base + (lane & 15);
		const int chunk = 4 * s + (lane >> 4);
		return *reinterpret_cast<const bf16x8*>(tile + swz_k(row, chunk));
	} else {
		const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
		const int krow = 32 * s + 8 * g + q;
		const int col = base + 4 * p;
		const int off = swz_mn(krow, col >> 3) + ((p & 1) << 3);
		typedef s16x4 __attribute__((address_space(3))) * lds_ptr_t;
		s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + off));
		s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(tile + off + 4 * 256));
		union {
			s16x4 h[2];
			bf16x8 v;
		} u;
		u.h[0] = lo;
		u.h[1] = hi;
		return u.v;
	}
}

// Global -> register staging of one operand tile: 4 chunks of 16 bytes per thread.
// rows_total/cols_total bound the operand; OOB chunks read as zero (K tail) or a clamped row.
template <int LAYOUT, bool PATCH>
__device__ __forceinline__ void load_tile(const GemmDev& g, const bf16_t* __restrict__ base, int64_t ld, int64_t mn0, int64_t mn_total, int k0, int k_end,
	uint4 (&regs)[4], int tid) {
	if (LAYOUT == VITK_K_MAJOR) {
		const int c = tid & 7;
		const int k = k0 + 8 * c;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int r = (tid >> 3) + 32 * i;
			int64_t row = mn0 + r;
			if (row >= mn_total) row = mn_total - 1;
			uint4 v = make_uint4(0, 0, 0, 0);
			if (k < k_end) {
				const bf16_t* p = PATCH ? base + patch_offset(g, row, k) : base + row * ld + k;
				v = *reinterpret_cast<const uint4*>(p);
			}
			regs[i] = v;
		}
	} else {
		const int c = tid & 15;
		const int64_t col = mn0 + 8 * c;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int r = (tid >> 4) + 16 * i;
			const int k = k0 + r;
			uint4 v = make_uint4(0, 0, 0, 0);
			if (k < k_end && col < mn_total) v = *reinterpret_cast<const uint4*>(base + (int64_t)k * ld + col);
			regs[i] = v;
		}
	}
}

template <int LAYOUT>
__device__ __forceinline__ void store_tile(char* tile, const uint4 (&regs)[4], int tid) {
	if (LAYOUT == VITK_K_MAJOR) {
		const int c = tid & 7;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int r = (tid >> 3) + 32 * i;
			*reinterpret_cast<uint4*>(tile + swz_k(r, c)) = regs[i];
		}
	} else {
		const int c = tid & 15;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int r = (tid >> 4) + 16 * i;
			*reinterpret_cast<uint4*>(tile + swz_mn(r, c)) = regs[i];
		}
	}
}

struct BfArgs {
	GemmDev g;
	int tiles_m, tiles_n, splits, k_per_split;  // k_per_split is a multiple of BK
	int vec4;                                    // epilogue may use 4-wide accesses
	int dma;                                     // K % 64 == 0: operand tiles by LDS-DMA; 1: K-major x K-major, 2: MN-major x MN-major, 3: any layouts, four stages
	int cs_turns;                                // workgroups (tile columns) that take turns, K tile by K tile, at the column sums of their A tile
	float* partials;                             // K split through scratch: [split][tile][128 x 128] partial tiles (else float atomics into C)
	float* cs_partials;                          // [split][M] partial column sums of A (with partials and a_colsum)
	float* a_colsum;                             // MN-major A: a_colsum[m] += sum_k A(m, k), formed from the staged A tiles
};

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

// LDS-DMA of one 128 x 64 K-major operand tile (global_load_lds_dwordx4, no register staging and no ds_write): 16
// pieces of 8 rows x 128 bytes, 4 per wave.  A wave-instruction writes its 64 x 16 bytes linearly, so the image's
// chunk swizzle is applied to each lane's SOURCE address; rows past the end re-read the last row (never stored).
__device__ __forceinline__ void dma_tile_k(char* tile, const bf16_t* __restrict__ base, int64_t ld, int64_t mn0, int64_t mn_total, int k0, int wave, int lane) {
	const int r = lane >> 3, slot = lane & 7;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int piece = 4 * wave + j;
		int64_t row = mn0 + 8 * piece + r;
		if (row > mn_total - 1) row = mn_total - 1;
		const bf16_t* src = base + row * ld + k0 + 8 * (slot ^ r);
		__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(tile + piece * 1024), 16, 0, 0);
	}
}

// The same for one [64 k][128 columns] MN-major operand tile (256-byte rows): 16 pieces of 4 k-rows, wave w takes pieces
// 2 w, 2 w + 1, 2 w + 8, 2 w + 9, for all of which bit 3 of the row index is (w & 1): the unit swizzle of swz_mn is then one
// loop-invariant lane offset.  Lane l lands at row l >> 4, physical 16-byte chunk l & 15 of its piece, so it fetches the
// logical chunk that swz_mn puts there.  Columns past the operand's last re-read its last 8 (they only reach output rows /
// columns that are never stored).
__device__ __forceinline__ void dma_tile_mn(char* tile, const bf16_t* __restrict__ base, int64_t ld, int64_t mn0, int64_t mn_total, int k0, int wave, int lane) {
	const int prow = lane >> 4, pchunk = lane & 15;
	const int lchunk = (((pchunk >> 1) ^ (prow | ((wave & 1) << 2))) << 1) | (pchunk & 1);
	int64_t col = mn0 + 8 * lchunk;
	if (col > mn_total - 8) col = mn_total - 8;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int piece = 2 * wave + (j & 1) + 8 * (j >> 1);
		const bf16_t* src = base + (int64_t)(k0 + 4 * piece + prow) * ld + col;
		__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(tile + piece * 1024), 16, 0, 0);
	}
}

__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(lds_void_p)(void*)p; }
// (inline asm: written as the builtin, the compiler waits vmcnt(0) in front of a transposing read, i.e. for the next
// tile's LDS-DMA that was just issued)
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

template <typename TOUT>
__device__ __forceinline__ void store_frag(const BfArgs& a, int64_t m, int64_t nb, f32x4 v) {
	const GemmDev& g = a.g;
	if (a.vec4 && nb + 3 < g.N) {
		const float f[4] = {v[0], v[1], v[2], v[3]};
		epi_store4<bf16_t, TOUT>(g, m, nb, f);
	} else {
		if (nb + 0 < g.N) epi_store1<bf16_t, TOUT>(g, m, nb + 0, v[0]);
		if (nb + 1 < g.N) epi_store1<bf16_t, TOUT>(g, m, nb + 1, v[1]);
		if (nb + 2 < g.N) epi_store1<bf16_t, TOUT>(g, m, nb + 2, v[2]);
		if (nb + 3 < g.N) epi_store1<bf16_t, TOUT>(g, m, nb + 3, v[3]);
	}
}

// One 128 x 128 output tile over one K slice: workgroup `wg_in` of the problem's tiles_m * tiles_n, K slice `split`.
// PROD: the four-stage form in a 512-thread workgroup whose waves 4 .. 7 only issue the LDS-DMA requests (and leave before
// the epilogue), so that waves 0 .. 3 never pay for them.
template <int ALAYOUT, int BLAYOUT, bool PATCH, typename TOUT, bool XCD_REMAP = true, bool PROD = false>
__device__ __forceinline__ void gemm_bf16_tile(BfArgs& a, char* smem, const int wg_in, const int split) {
	gemm_dev_enter(a.g);
	const GemmDev& g = a.g;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

	// XCD-aware, bijective remap of the linear workgroup id, then grouped (8 tile-rows) raster so that
	// the workgroups sharing an XCD's L2 work on neighbouring tiles.
	const int nwg = a.tiles_m * a.tiles_n;
	int wg = wg_in;
	if (XCD_REMAP) {
		const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, idx = wg >> 3;
		wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
	}
	constexpr int GM = 8;
	const int per_group = GM * a.tiles_n;
	const int group = wg / per_group;
	const int first_m = group * GM;
	const int gsz = min(a.tiles_m - first_m, GM);
	const int tm = first_m + (wg % per_group) % gsz;
	const int tn = (wg % per_group) / gsz;
	const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

	const int k_begin = split * a.k_per_split;
	const int k_end = min((int)g.K, k_begin + a.k_per_split);
	const int nt = (k_end - k_begin + BK - 1) / BK;

	const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
	const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);

	f32x4 acc[4][4];
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

	auto mma_tile = [&](const char* sa, const char* sb) {
#pragma unroll
		for (int s = 0; s < 2; ++s) {
			bf16x8 xf[4], wf[4];
#pragma unroll
			for (int i = 0; i < 4; ++i) xf[i] = read_frag<ALAYOUT>(sa, wm + 16 * i, s, lane);
#pragma unroll
			for (int j = 0; j < 4; ++j) wf[j] = read_frag<BLAYOUT>(sb, wn + 16 * j, s, lane);
#pragma unroll
			for (int i = 0; i < 4; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
		}
	};

	// Column sums of an MN-major A (the bias gradient next to a weight gradient): a thread stages 8 fixed columns of
	// 4 k rows per K tile through its registers anyway, so it adds them up on the way; the workgroups of the first tile
	// column do it (all tile columns stage the same A).  Out-of-range pieces were read as zeros.
	bool do_colsum = false;
	if constexpr (ALAYOUT == VITK_MN_MAJOR) do_colsum = a.a_colsum != nullptr && tn < a.cs_turns;
	float csum[8];
#pragma unroll
	for (int e = 0; e < 8; ++e) csum[e] = 0.f;
	auto add_colsum = [&](const uint4 (&r)[4]) {
		if constexpr (ALAYOUT == VITK_MN_MAJOR) {
			if (do_colsum) {
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const bf16x8 v = __builtin_bit_cast(bf16x8, r[i]);
#pragma unroll
					for (int e = 0; e < 8; ++e) csum[e] += (float)v[e];
				}
			}
		}
	};

	bool by_dma = false, by_dma_mn = false;
	if constexpr (ALAYOUT == VITK_K_MAJOR && BLAYOUT == VITK_K_MAJOR && !PATCH) by_dma = a.dma == 1;
	if constexpr (ALAYOUT == VITK_MN_MAJOR && BLAYOUT == VITK_MN_MAJOR && !PATCH) by_dma_mn = a.dma == 2;
	float csum1 = 0.f;  // by_dma_mn: column tid & 127 of the A tile over k-rows 32 (tid >> 7) .. + 31, of the K tiles that are this workgroup's turn
	const bool by_deep = PROD ? true : (!PATCH && a.dma == 3);
	if (by_deep) {
		// Few workgroups (at most one per CU) over a K of several tiles: with two stages a K tile costs a whole memory round
		// trip, because the only request in flight is the next tile's.  Four stages, three tiles requested ahead, counted
		// waits: tile t is waited for with the requests of up to two newer tiles (8 per wave and tile) still in flight.
		// Either layout per operand; every LDS read is inline asm (the compiler would wait vmcnt(0) in front of reads it
		// can see).  The barrier of step t also tells that every wave is done with stage (t - 1) % 4, which the request
		// issued right behind it refills.  PROD: a request costs the wave that issues it about 100 cycles, 800 per K tile --
		// as much as the tile's fragment reads and MFMAs -- so four more waves do nothing else: they wait for their own
		// requests, meet the others at the barrier (which publishes the tile) and request three tiles ahead.
		constexpr int ST = DEEP_STAGES;
		const bool requests = !PROD || wave >= 4, computes = !PROD || wave < 4;  // wave-uniform
		const int pw = wave & 3;
		const uint32_t lds0 = lds_addr(smem);
		const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
		auto lane_k = [&](int base, int s) { return (uint32_t)swz_k(base + (lane & 15), 4 * s + fg); };  // + 2048 per fragment
		auto lane_mn = [&](int base) { return (uint32_t)(swz_mn(8 * fg + fq, (base >> 3) + (fp >> 1)) + ((fp & 1) << 3)); };  // ^ (fragment << 5)
		const uint32_t a_k0 = lane_k(wm, 0), a_k1 = lane_k(wm, 1), b_k0 = lane_k(wn, 0), b_k1 = lane_k(wn, 1);
		const uint32_t a_mn = lane_mn(wm), b_mn = lane_mn(wn);
		auto issue = [&](int t) {  // K tile t of this slice into stage t % ST
			char* da = smem + (t % ST) * STAGE_BYTES;
			const int k0 = k_begin + t * BK;
			if constexpr (ALAYOUT == VITK_K_MAJOR) dma_tile_k(da, A, g.lda, m0, g.M, k0, pw, lane);
			else dma_tile_mn(da, A, g.lda, m0, g.M, k0, pw, lane);
			if constexpr (BLAYOUT == VITK_K_MAJOR) dma_tile_k(da + TILE_BYTES, B, g.ldb, n0, g.N, k0, pw, lane);
			else dma_tile_mn(da + TILE_BYTES, B, g.ldb, n0, g.N, k0, pw, lane);
		};
		if (requests)
			for (int t = 0; t < ST - 1 && t < nt; ++t) issue(t);
		for (int t = 0; t < nt; ++t) {
			if (requests) {
				const int newer = min(nt - 1, t + ST - 2) - t;  // tiles requested behind tile t
				if (newer >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
				else if (newer == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
				else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
			__builtin_amdgcn_s_barrier();
			if (requests && t + ST - 1 < nt) issue(t + ST - 1);
			if (!computes) continue;
			const uint32_t sa = lds0 + (t % ST) * STAGE_BYTES, sb = sa + TILE_BYTES;
			u32x4 ka[4][2], kb[4][2];            // K-major fragments [fragment][k half]
			s16x4 ta[4][2][2], tb[4][2][2];      // MN-major fragments [fragment][k half][read]
			auto read_half = [&](int s) {
				if constexpr (BLAYOUT == VITK_K_MAJOR) {
#pragma unroll
					for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(kb[j][s]) : "v"(sb + (s ? b_k1 : b_k0) + 2048 * j));
				} else {
#pragma unroll
					for (int j = 0; j < 4; ++j) {
						asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(tb[j][s][0]) : "v"(sb + (b_mn ^ (j << 5)) + 8192 * s));
						asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(tb[j][s][1]) : "v"(sb + (b_mn ^ (j << 5)) + 8192 * s));
					}
				}
				if constexpr (ALAYOUT == VITK_K_MAJOR) {
#pragma unroll
					for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(ka[i][s]) : "v"(sa + (s ? a_k1 : a_k0) + 2048 * i));
				} else {
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ta[i][s][0]) : "v"(sa + (a_mn ^ (i << 5)) + 8192 * s));
						asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(ta[i][s][1]) : "v"(sa + (a_mn ^ (i << 5)) + 8192 * s));
					}
				}
			};
			auto landed = [&](int s) {  // ties the registers of k half s to the wait in front of this statement
				if constexpr (BLAYOUT == VITK_K_MAJOR) asm volatile("" : "+v"(kb[0][s]), "+v"(kb[1][s]), "+v"(kb[2][s]), "+v"(kb[3][s])::"memory");
				else asm volatile("" : "+v"(tb[0][s][0]), "+v"(tb[0][s][1]), "+v"(tb[1][s][0]), "+v"(tb[1][s][1]), "+v"(tb[2][s][0]), "+v"(tb[2][s][1]), "+v"(tb[3][s][0]), "+v"(tb[3][s][1])::"memory");
				if constexpr (ALAYOUT == VITK_K_MAJOR) asm volatile("" : "+v"(ka[0][s]), "+v"(ka[1][s]), "+v"(ka[2][s]), "+v"(ka[3][s])::"memory");
				else asm volatile("" : "+v"(ta[0][s][0]), "+v"(ta[0][s][1]), "+v"(ta[1][s][0]), "+v"(ta[1][s][1]), "+v"(ta[2][s][0]), "+v"(ta[2][s][1]), "+v"(ta[3][s][0]), "+v"(ta[3][s][1])::"memory");
			};
			auto mma_half = [&](int s) {
#pragma unroll
				for (int i = 0; i < 4; ++i)
#pragma unroll
					for (int j = 0; j < 4; ++j) {
						bf16x8 xf, wf;
						if constexpr (ALAYOUT == VITK_K_MAJOR) xf = __builtin_bit_cast(bf16x8, ka[i][s]);
						else xf = join8(ta[i][s][0], ta[i][s][1]);
						if constexpr (BLAYOUT == VITK_K_MAJOR) wf = __builtin_bit_cast(bf16x8, kb[j][s]);
						else wf = join8(tb[j][s][0], tb[j][s][1]);
						acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[i][j], 0, 0, 0);
					}
			};
			// LDS reads return in order: the second half's reads (8 ... 16 of them) may still be in flight while the first
			// half's MFMAs run
			constexpr int HALF_READS = (ALAYOUT == VITK_K_MAJOR ? 4 : 8) + (BLAYOUT == VITK_K_MAJOR ? 4 : 8);
			read_half(0);
			read_half(1);
			if constexpr (HALF_READS == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
			else if constexpr (HALF_READS == 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
			else asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
			landed(0);
			mma_half(0);
			__builtin_amdgcn_sched_barrier(0);
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			landed(1);
			mma_half(1);
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (PROD && !computes) return;  // (a barrier waits for the waves that are still there)
		__syncthreads();  // the epilogue image goes over the stages
	} else if (by_dma_mn) {
		// as below (one barrier per K tile), with both MN-major tiles staged by LDS-DMA and every LDS read issued as inline asm
		const uint32_t lds0 = lds_addr(smem);
		const int fg = lane >> 4, fq = (lane >> 2) & 3, fp = lane & 3;
		const uint32_t a_lane = (uint32_t)(swz_mn(8 * fg + fq, (wm >> 3) + (fp >> 1)) + ((fp & 1) << 3));
		const uint32_t b_lane = (uint32_t)(swz_mn(8 * fg + fq, (wn >> 3) + (fp >> 1)) + ((fp & 1) << 3));
		const int cs_c = tid & 127, cs_r = tid >> 7;
		dma_tile_mn(smem, A, g.lda, m0, g.M, k_begin, wave, lane);
		dma_tile_mn(smem + TILE_BYTES, B, g.ldb, n0, g.N, k_begin, wave, lane);
		for (int t = 0; t < nt; ++t) {
			__syncthreads();
			const uint32_t sa = lds0 + (t & 1) * STAGE_BYTES, sb = sa + TILE_BYTES;
			if (t + 1 < nt) {
				char* da = smem + ((t + 1) & 1) * STAGE_BYTES;
				const int k0 = k_begin + (t + 1) * BK;
				dma_tile_mn(da, A, g.lda, m0, g.M, k0, wave, lane);
				dma_tile_mn(da + TILE_BYTES, B, g.ldb, n0, g.N, k0, wave, lane);
			}
			if (do_colsum && (k_begin / BK + t) % a.cs_turns == tn) {
#pragma unroll
				for (int b4 = 0; b4 < 4; ++b4) {
					uint32_t v[8];
#pragma unroll
					for (int e = 0; e < 8; ++e) {
						const int kr = 32 * cs_r + 8 * b4 + e;
						const uint32_t ad = sa + (uint32_t)(swz_mn(kr, cs_c >> 3) + (cs_c & 7) * 2);
						asm volatile("ds_read_u16 %0, %1" : "=v"(v[e]) : "v"(ad));
					}
					asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])::"memory");
#pragma unroll
					for (int e = 0; e < 8; ++e) csum1 += __builtin_bit_cast(float, v[e] << 16);
				}
			}
			s16x4 xa[4][2][2], xb[4][2][2];  // [fragment][k half][read]
			// all 32 reads up front, k half 0 first; LDS reads return in order and the counter holds 15 at most, so "at most 15
			// outstanding" means the first 17 have landed
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				lds_read_tr(xb[j][0][0], sb + (b_lane ^ (j << 5)), 0);
				lds_read_tr(xb[j][0][1], sb + (b_lane ^ (j << 5)), 1024);
			}
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				lds_read_tr(xa[i][0][0], sa + (a_lane ^ (i << 5)), 0);
				lds_read_tr(xa[i][0][1], sa + (a_lane ^ (i << 5)), 1024);
			}
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				lds_read_tr(xb[j][1][0], sb + (b_lane ^ (j << 5)), 8192);
				lds_read_tr(xb[j][1][1], sb + (b_lane ^ (j << 5)), 8192 + 1024);
			}
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				lds_read_tr(xa[i][1][0], sa + (a_lane ^ (i << 5)), 8192);
				lds_read_tr(xa[i][1][1], sa + (a_lane ^ (i << 5)), 8192 + 1024);
			}
			asm volatile("s_waitcnt lgkmcnt(15)"
						 : "+v"(xb[0][0][0]), "+v"(xb[0][0][1]), "+v"(xb[1][0][0]), "+v"(xb[1][0][1]), "+v"(xb[2][0][0]), "+v"(xb[2][0][1]), "+v"(xb[3][0][0]), "+v"(xb[3][0][1]),
						   "+v"(xa[0][0][0]), "+v"(xa[0][0][1]), "+v"(xa[1][0][0]), "+v"(xa[1][0][1]), "+v"(xa[2][0][0]), "+v"(xa[2][0][1]), "+v"(xa[3][0][0]), "+v"(xa[3][0][1])
						 :
						 : "memory");
#pragma unroll
			for (int i = 0; i < 4; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j)
					acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(xb[j][0][0], xb[j][0][1]), join8(xa[i][0][0], xa[i][0][1]), acc[i][j], 0, 0, 0);
			__builtin_amdgcn_sched_barrier(0);  // (the first half's MFMAs stay in front of the wait for the second half's reads)
			asm volatile("s_waitcnt lgkmcnt(0)"
						 : "+v"(xb[0][1][0]), "+v"(xb[0][1][1]), "+v"(xb[1][1][0]), "+v"(xb[1][1][1]), "+v"(xb[2][1][0]), "+v"(xb[2][1][1]), "+v"(xb[3][1][0]), "+v"(xb[3][1][1]),
						   "+v"(xa[0][1][0]), "+v"(xa[0][1][1]), "+v"(xa[1][1][0]), "+v"(xa[1][1][1]), "+v"(xa[2][1][0]), "+v"(xa[2][1][1]), "+v"(xa[3][1][0]), "+v"(xa[3][1][1])
						 :
						 : "memory");
#pragma unroll
			for (int i = 0; i < 4; ++i)
#pragma unroll
				for (int j = 0; j < 4; ++j)
					acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join8(xb[j][1][0], xb[j][1][1]), join8(xa[i][1][0], xa[i][1][1]), acc[i][j], 0, 0, 0);
		}
		__syncthreads();  // the epilogue image goes over the stages
	} else if (by_dma) {
		// one barrier per K tile: it retires this tile's DMA (the fence in front of it drains vmcnt) and tells that every
		// wave is done with the other stage, which the next tile's DMA then overwrites
		dma_tile_k(smem, A, g.lda, m0, g.M, k_begin, wave, lane);
		dma_tile_k(smem + TILE_BYTES, B, g.ldb, n0, g.N, k_begin, wave, lane);
		for (int t = 0; t < nt; ++t) {
			__syncthreads();
			const char* sa = smem + (t & 1) * STAGE_BYTES;
			if (t + 1 < nt) {
				char* da = smem + ((t + 1) & 1) * STAGE_BYTES;
				const int k0 = k_begin + (t + 1) * BK;
				dma_tile_k(da, A, g.lda, m0, g.M, k0, wave, lane);
				dma_tile_k(da + TILE_BYTES, B, g.ldb, n0, g.N, k0, wave, lane);
			}
			mma_tile(sa, sa + TILE_BYTES);
		}
		__syncthreads();  // the epilogue image goes over the stages
	} else {
		uint4 ra[4], rb[4];
		load_tile<ALAYOUT, PATCH>(g, A, g.lda, m0, g.M, k_begin, k_end, ra, tid);
		load_tile<BLAYOUT, false>(g, B, g.ldb, n0, g.N, k_begin, k_end, rb, tid);
		add_colsum(ra);
		store_tile<ALAYOUT>(smem, ra, tid);
		store_tile<BLAYOUT>(smem + TILE_BYTES, rb, tid);
		__syncthreads();

		for (int t = 0; t < nt; ++t) {
			const char* sa = smem + (t & 1) * STAGE_BYTES;
			const char* sb = sa + TILE_BYTES;
			const bool more = (t + 1 < nt);
			if (more) {
				const int k0 = k_begin + (t + 1) * BK;
				load_tile<ALAYOUT, PATCH>(g, A, g.lda, m0, g.M, k0, k_end, ra, tid);
				load_tile<BLAYOUT, false>(g, B, g.ldb, n0, g.N, k0, k_end, rb, tid);
			}
			mma_tile(sa, sb);
			if (more) {
				add_colsum(ra);
				char* da = smem + ((t + 1) & 1) * STAGE_BYTES;
				store_tile<ALAYOUT>(da, ra, tid);
				store_tile<BLAYOUT>(da + TILE_BYTES, rb, tid);
			}
			__syncthreads();
		}
	}

	// Epilogue through LDS: every lane drops its 16 accumulator fragments into a [128][128] float image of the
	// output tile (rows padded by 16 bytes: the 8 lanes of a ds_write_b128 group hold 8 different rows), then the
	// workgroup walks the image row-major, 4 consecutive columns per lane, so that the fused epilogue reads its
	// residual / pre-activation operand and writes C in whole 128-byte lines.  The epilogue code exists once, in a
	// runtime loop, and every index into acc[][] is a compile-time constant (a runtime index would move acc to scratch).
	// acc[i][j][r]: m = wm + 16 i + (lane & 15), n = wn + 16 j + 4 (lane >> 4) + r
	float* ct = reinterpret_cast<float*>(smem);
	if constexpr (ALAYOUT == VITK_MN_MAJOR) {
		if (do_colsum) {  // workgroup-uniform: the 16 (2) row groups of a column combine through LDS (the stages are free now)
			if (by_dma_mn) {
				ct[tid] = csum1;
			} else {
#pragma unroll
				for (int e = 0; e < 8; ++e) ct[(tid >> 4) * 128 + 8 * (tid & 15) + e] = csum[e];
			}
			__syncthreads();
			if (tid < 128 && m0 + tid < g.M) {
				float sum = 0.f;
				const int groups = by_dma_mn ? 2 : 16;
				for (int r = 0; r < groups; ++r) sum += ct[r * 128 + tid];
				if (a.cs_partials != nullptr) a.cs_partials[((int64_t)split * a.cs_turns + tn) * g.M + m0 + tid] = sum;
				else if (a.splits > 1) atomicAdd(a.a_colsum + m0 + tid, sum);
				else a.a_colsum[m0 + tid] += sum;  // the only workgroup that owns these rows
			}
			__syncthreads();
		}
	}
#pragma unroll
	for (int i = 0; i < 4; ++i)
#pragma unroll
		for (int j = 0; j < 4; ++j)
			*reinterpret_cast<f32x4*>(ct + (wm + 16 * i + (lane & 15)) * CT_LD + wn + 16 * j + 4 * (lane >> 4)) = acc[i][j];
	__syncthreads();
	if (a.splits > 1 && a.partials != nullptr) {
		// K-split partial tile into scratch, row-major 128 x 128 (512 contiguous bytes per row); split_reduce128_kernel
		// adds the slices up: deterministic, and no float atomics (atomics of many workgroups on one address run one
		// after the other at the memory side, about 0.2 us each)
		float* dst = a.partials + ((int64_t)split * nwg + (int64_t)tm * a.tiles_n + tn) * (BM * BN);
		for (int it = 0; it < (BM * BN / 4) / 256; ++it) {
			const int v = it * 256 + tid;
			const int row = v >> 5, c4 = v & 31;
			*reinterpret_cast<f32x4*>(dst + row * BN + 4 * c4) = *reinterpret_cast<const f32x4*>(ct + row * CT_LD + 4 * c4);
		}
		return;
	}
	if (a.splits > 1) {
		// split-K partial sums: float atomics, one dword per lane on consecutive addresses
		float* cbase = reinterpret_cast<float*>(g.C);
		for (int it = 0; it < (BM * BN) / 256; ++it) {
			const int v = it * 256 + tid;
			const int row = v >> 7, col = v & 127;
			const int64_t m = m0 + row, n = n0 + col;
			if (m < g.M && n < g.N) atomicAdd(cbase + m * g.ldc + n, ct[row * CT_LD + col]);
		}
		return;
	}
	for (int it = 0; it < (BM * BN / 4) / 256; ++it) {
		const int v = it * 256 + tid;
		const int row = v >> 5, c4 = v & 31;
		const int64_t m = m0 + row, nb = n0 + 4 * c4;
		if (m >= g.M || nb >= g.N) continue;
		const f32x4 val = *reinterpret_cast<const f32x4*>(ct + row * CT_LD + 4 * c4);
		store_frag<TOUT>(a, m, nb, val);
	}
}

template <int ALAYOUT, int BLAYOUT, bool PATCH, typename TOUT>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(BfArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	gemm_bf16_tile<ALAYOUT, BLAYOUT, PATCH, TOUT>(a, smem, blockIdx.x, blockIdx.y);
}

template <int ALAYOUT, int BLAYOUT, typename TOUT>
__global__ __launch_bounds__(512, 1) void gemm_bf16_deep_kernel(BfArgs a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	gemm_bf16_tile<ALAYOUT, BLAYOUT, false, TOUT, true, true>(a, smem, blockIdx.x, blockIdx.y);
}

// Several independent weight-gradient products (MN-major x MN-major, float32 result, K split through scratch) in one
// launch: a block of a small model has four of them, each too small to fill the chip without cutting K so finely that
// the partial tiles cost as much as the product.  Workgroups [first[i], first[i + 1]) belong to problem i, K slice by
// K slice.
constexpr int GROUP_MAX = 8;
struct BfGroupArgs {
	BfArgs p[GROUP_MAX];
	int first[GROUP_MAX + 1];
	int n;
};
static_assert(sizeof(BfGroupArgs) <= 4096, "kernel argument block");

__global__ __launch_bounds__(256, 2) void gemm_bf16_group_kernel(BfGroupArgs ga) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	// consecutive workgroup ids go to the 8 XCDs in turn: XCD x takes the x-th eighth of the list (problem, K slice, tile),
	// so that the tiles of one K slice -- which share its operand rows -- run side by side behind one L2
	const int total = ga.first[ga.n];
	int v = blockIdx.x;
	{
		const int q = total >> 3, r = total & 7, xcd = v & 7, idx = v >> 3;
		v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
	}
	int gi = 0;
	while (gi + 1 < ga.n && v >= ga.first[gi + 1]) ++gi;
	BfArgs a = ga.p[gi];
	const int local = v - ga.first[gi], nwg = a.tiles_m * a.tiles_n;
	gemm_bf16_tile<VITK_MN_MAJOR, VITK_MN_MAJOR, false, float, false>(a, smem, local % nwg, local / nwg);
}

// C (+)= sum over the K slices of the partial tiles (block row `ty` < ntile),
// and a_colsum += sum over the slices of the partial column sums (ty == ntile).
struct ReduceItem {
	const float* partials;
	float* C;
	int64_t M, N, ldc;
	int tiles_n, ntile, splits, accumulate;
	const float* cs_partials;  // [cs_slots][M]
	float* a_colsum;
	int cs_slots;
};

__device__ __forceinline__ void split_reduce128_rows(const ReduceItem& r, const int ty) {
	if (ty == r.ntile) {
		if (r.cs_partials == nullptr) return;
		for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < r.M; m += (int64_t)gridDim.x * 256) {
			float sum = 0.f;
			for (int s0 = 0; s0 < r.cs_slots; s0 += 16) {
				float t[16];
#pragma unroll
				for (int u = 0; u < 16; ++u) t[u] = s0 + u < r.cs_slots ? r.cs_partials[(int64_t)(s0 + u) * r.M + m] : 0.f;
#pragma unroll
				for (int u = 0; u < 16; ++u) sum += t[u];
			}
			r.a_colsum[m] += sum;
		}
		return;
	}
	// one thread per output element: four times the loads in flight of a 16-byte-per-thread walk, which is what this
	// short, latency-bound kernel (tens of slices of a few tiles) needs; a wave still reads 256 contiguous bytes
	const int tile = ty;
	const int q = blockIdx.x * 256 + threadIdx.x;
	const int row = q >> 7, col = q & 127;
	const int64_t m = (int64_t)(tile / r.tiles_n) * BM + row;
	const int64_t n = (int64_t)(tile % r.tiles_n) * BN + col;
	if (m >= r.M || n >= r.N) return;
	const float* src = r.partials + (int64_t)tile * (BM * BN) + q;
	const int64_t step = (int64_t)r.ntile * (BM * BN);
	// 16 slices requested before the first is used: the loop is a chain of memory round trips otherwise
	float acc16[16];
#pragma unroll
	for (int u = 0; u < 16; ++u) acc16[u] = 0.f;
	int s = 0;
	for (; s + 15 < r.splits; s += 16) {
		float t[16];
#pragma unroll
		for (int u = 0; u < 16; ++u) t[u] = src[(int64_t)(s + u) * step];
#pragma unroll
		for (int u = 0; u < 16; ++u) acc16[u] += t[u];
	}
	{
		float t[16];
#pragma unroll
		for (int u = 0; u < 16; ++u) t[u] = s + u < r.splits ? src[(int64_t)(s + u) * step] : 0.f;
#pragma unroll
		for (int u = 0; u < 16; ++u) acc16[u] += t[u];
	}
#pragma unroll
	for (int w = 8; w > 0; w >>= 1)
#pragma unroll
		for (int u = 0; u < w; ++u) acc16[u] += acc16[u + w];
	const float v = acc16[0];
	float* c = r.C + m * r.ldc + n;
	*c = r.accumulate ? *c + v : v;
}

__global__ __launch_bounds__(256) void split_reduce128_kernel(ReduceItem r) { split_reduce128_rows(r, blockIdx.y); }

// the same for the problems of a grouped launch: block rows [first[i], first[i + 1]) are problem i's ntile + 1 rows
struct ReduceGroup {
	ReduceItem p[GROUP_MAX];
	int first[GROUP_MAX + 1];
	int n;
};

__global__ __launch_bounds__(256) void split_reduce128_group_kernel(ReduceGroup rg) {
	int gi = 0;
	while (gi + 1 < rg.n && (int)blockIdx.y >= rg.first[gi + 1]) ++gi;
	const ReduceItem r = rg.p[gi];
	split_reduce128_rows(r, (int)blockIdx.y - rg.first[gi]);
}

template <int AL, int BL, typename TOUT>
int launch_deep(const BfArgs& a, dim3 grid, hipStream_t st) {
	static bool attr_set = false;
	if (!attr_set) {
		hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_deep_kernel<AL, BL, TOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, DEEP_LDS_BYTES);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr_set = true;
	}
	vitk_last_gemm_kernel = "gemm_bf16_deep_kernel";
	hipLaunchKernelGGL((gemm_bf16_deep_kernel<AL, BL, TOUT>), grid, dim3(512), DEEP_LDS_BYTES, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

template <int AL, int BL, bool PATCH>
int launch(const BfArgs& a, int out_dtype, dim3 grid, hipStream_t st) {
	static const bool no_prod = getenv("VITK_NO_DEEP128_PRODUCERS") != nullptr;  // A-B switch
	if constexpr (!PATCH) {
		if (a.dma == 3 && !no_prod) return out_dtype == VITK_BF16 ? launch_deep<AL, BL, bf16_t>(a, grid, st) : launch_deep<AL, BL, float>(a, grid, st);
	}
	const size_t lds = a.dma == 3 ? DEEP_LDS_BYTES : LDS_BYTES;
	static bool attr_set[2] = {false, false};  // per template instantiation (function-local static)
	const int oi = out_dtype == VITK_BF16 ? 0 : 1;
	if (!attr_set[oi]) {
		hipError_t e = oi == 0 ? hipFuncSetAttribute((const void*)gemm_bf16_kernel<AL, BL, PATCH, bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, DEEP_LDS_BYTES)
							   : hipFuncSetAttribute((const void*)gemm_bf16_kernel<AL, BL, PATCH, float>, hipFuncAttributeMaxDynamicSharedMemorySize, DEEP_LDS_BYTES);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr_set[oi] = true;
	}
	if (out_dtype == VITK_BF16)
		hipLaunchKernelGGL((gemm_bf16_kernel<AL, BL, PATCH, bf16_t>), grid, dim3(256), lds, st, a);
	else
		hipLaunchKernelGGL((gemm_bf16_kernel<AL, BL, PATCH, float>), grid, dim3(256), lds, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

}  // namespace

bool vitk_gemm_bf16_ok(const vitk_gemm_desc& d) {
	if (d.in_dtype != VITK_BF16) return false;
	if (d.K % 8 != 0) return false;
	if (!vitk_aligned(d.A, 16) || !vitk_aligned(d.B, 16)) return false;
	if (d.patch_grid > 0) {
		const int p = d.img_hw / d.patch_grid;
		if (p % 8 != 0 || d.a_layout != VITK_K_MAJOR) return false;
	} else {
		if (d.lda % 8 != 0) return false;
		if (d.a_layout == VITK_MN_MAJOR && d.M % 8 != 0) return false;
	}
	if (d.ldb % 8 != 0) return false;
	if (d.b_layout == VITK_MN_MAJOR && d.N % 8 != 0) return false;
	// tiny problems: the generic kernel's 64x64 tiles waste less
	if ((d.M * d.N < 64 * 64 && d.K < 4096) || d.K < 32) return false;  // (a tiny output over a very long K -- the weight gradient of a narrow convolution -- still wants the K-split MFMA kernel)
	if (vitk_cdiv(d.M, BM) * vitk_cdiv(d.N, BN) > (1ll << 30)) return false;
	return true;
}

int vitk_gemm_bf16(const vitk_gemm_desc& d, hipStream_t st, bool* colsum_done) {
	static const bool big_tiles = getenv("VITK_NO_BIG_TILE") == nullptr;  // debugging / A-B switch
	*colsum_done = false;
	// Problems whose 256x256 tiles leave the last round of the persistent kernel mostly empty (N = 768 at 50,432 rows: 591
	// tiles = 2.3 rounds of 256) run on the 128x256-tile kernel, which balances the tail at half the granularity.
	// Few rows (CvT's last stage: 4096 tokens x 384 ... 1536 features): 256 x 256 tiles would occupy a fraction of the CUs with
	// one workgroup each and nothing to split; four times as many 128 x 128 tiles fill the chip better.  (Weight gradients
	// split K instead and stay on the large tiles.)
	static const bool small_rule = getenv("VITK_NO_SMALL_TILE_RULE") == nullptr;  // A-B switch
	const int64_t t256n = vitk_cdiv(d.M, 256) * vitk_cdiv(d.N, 256);
	const bool starved = small_rule && d.out_dtype == VITK_BF16 && t256n < 128;  // (thresholds 64 ... 256 measured the same on the CvT step: 7.9 -> 6.6 ms)
	static const bool p8 = getenv("VITK_NO_P8") == nullptr;  // A-B switch
	static const bool force_t128 = getenv("VITK_FORCE_T128") != nullptr;  // diagnosis: every problem the 128x256-tile kernel takes runs on it
	if (force_t128 && vitk_gemm_bf16_t128_ok(d) && d.a_colsum == nullptr) return vitk_gemm_bf16_t128(d, st);
	// (at least 64 tiles' worth of output: measured on the 32 x 32 models, 32 / 64 / 100 tiles as the bound: CvT 4.09 / 4.04 / 4.04 ms,
	// ViT-Tiny 4.91 / 4.91 / 5.64 ms per step)
	if (p8 && big_tiles && d.a_colsum == nullptr && vitk_gemm_bf16_p8_ok(d) && d.M * d.N >= 64ll * 256 * 256) {
		// (Cutting a launch whose last round is mostly empty -- N = 768: 591 tiles = 2.31 rounds -- into whole rounds here plus
		// the remaining rows on the 128x256-tile kernel was measured on the ViT-B/16 step: 0.6 ms slower, so it is not done.)
		return vitk_gemm_bf16_p8(d, st);
	}
	static const bool t128 = getenv("VITK_NO_T128") == nullptr;  // A-B switch
	if (t128 && big_tiles && vitk_gemm_bf16_t128_ok(d) && d.a_colsum == nullptr) {
		const int64_t t256 = vitk_cdiv(d.M, 256) * vitk_cdiv(d.N, 256), t128n = vitk_cdiv(d.M, 128) * vitk_cdiv(d.N, 256);
		const int64_t r256 = vitk_cdiv(t256, 256), r128 = vitk_cdiv(t128n, 256);  // rounds in units of a CU's tile time / half of it
		if (t256 > 256 && r128 * 10 < r256 * 2 * 9) return vitk_gemm_bf16_t128(d, st);
	}
	if (big_tiles && !starved && vitk_gemm_bf16_256_ok(d)) {
		const int rc = vitk_gemm_bf16_256(d, st, colsum_done);
		if (rc != VITK_ERR_UNSUPPORTED) return rc;
	}
	return vitk_gemm_bf16_128(d, st, colsum_done);
}

// the fields of a 128 x 128 launch that do not depend on how K is cut
static bool fill128(const vitk_gemm_desc& d, BfArgs& a) {
	a.g = make_gemm_dev(d);
	a.tiles_m = (int)vitk_cdiv(d.M, BM);
	a.tiles_n = (int)vitk_cdiv(d.N, BN);
	auto al16 = [](const void* p) { return p == nullptr || vitk_aligned(p, 16); };
	a.vec4 = (d.N % 4 == 0) && (d.ldc % 4 == 0) && (d.ld_aux % 4 == 0) && (d.ld_out2 % 4 == 0) && al16(d.C) && al16(d.bias) && al16(d.gamma) && al16(d.pos) &&
		al16(d.aux) && al16(d.out2);
	// MN-major A tiles pass through registers, where their column sums (the bias gradient) are formed on the way
	static const bool no_fused_cs = getenv("VITK_NO_FUSED_COLSUM128") != nullptr;  // A-B switch
	const bool fuse_colsum = !no_fused_cs && d.a_colsum != nullptr && d.a_layout == VITK_MN_MAJOR && d.patch_grid == 0;
	if (fuse_colsum) a.a_colsum = d.a_colsum;
	static const bool no_dma = getenv("VITK_NO_DMA128") != nullptr;  // A-B switch
	a.dma = !no_dma && d.patch_grid == 0 && d.a_layout == VITK_K_MAJOR && d.b_layout == VITK_K_MAJOR && d.K % BK == 0 && d.M >= 1 && d.N >= 1;
	static const bool no_dma_mn = getenv("VITK_NO_DMA128_MN") != nullptr;  // A-B switch
	if (!no_dma_mn && d.patch_grid == 0 && d.a_layout == VITK_MN_MAJOR && d.b_layout == VITK_MN_MAJOR && d.K % BK == 0 && d.M >= 8 && d.N >= 8) a.dma = 2;
	a.cs_turns = 1;
	return fuse_colsum;
}

static bool splits_k_128(const vitk_gemm_desc& d) {  // the plain float32-output form (weight gradients): few output tiles, long K
	return d.out_dtype == VITK_F32 && d.epilogue == VITK_EPI_NONE && d.bias == nullptr && d.tokens_per_image == 0 &&
		vitk_cdiv(d.M, BM) * vitk_cdiv(d.N, BN) <= 256 && vitk_cdiv(d.K, BK) >= 16;
}

int vitk_gemm_bf16_128(const vitk_gemm_desc& d, hipStream_t st, bool* colsum_done) {
	vitk_last_gemm_kernel = "gemm_bf16_kernel";
	BfArgs a{};
	const bool fuse_colsum = fill128(d, a);
	const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
	const int64_t ktiles = vitk_cdiv(d.K, BK);
	a.splits = 1;
	if (splits_k_128(d)) {
		int64_t s = std::min<int64_t>(512 / tiles, ktiles / 8);  // two workgroups per CU
		a.splits = (int)std::max<int64_t>(1, std::min<int64_t>(s, 64));
	}
	const int64_t kt_per = vitk_cdiv(ktiles, a.splits);
	a.k_per_split = (int)(kt_per * BK);
	a.splits = (int)vitk_cdiv(ktiles, kt_per);
	if (a.splits > 1) {
		const int64_t tile_bytes = (int64_t)a.splits * tiles * BM * BN * (int64_t)sizeof(float);
		const int turns = fuse_colsum && a.dma == 2 ? a.tiles_n : 1;
		const int64_t need = tile_bytes + (fuse_colsum ? (int64_t)a.splits * turns * d.M * (int64_t)sizeof(float) : 0);
		if (d.workspace != nullptr && d.workspace_bytes >= need && vitk_aligned(d.workspace, 16)) {
			a.partials = reinterpret_cast<float*>(d.workspace);
			if (fuse_colsum) {
				a.cs_partials = a.partials + tile_bytes / (int64_t)sizeof(float);
				a.cs_turns = turns;
			}
		}
	}
	if (a.splits > 1 && a.partials == nullptr && !d.accumulate) {
		hipError_t e = vitk_zero_f32_2d(reinterpret_cast<float*>(d.C), d.M, d.N, d.ldc, st);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	}
	// at most one workgroup per CU and a K of several tiles per workgroup: the four-stage form (see the kernel)
	static const bool no_deep = getenv("VITK_NO_DEEP128") != nullptr;  // A-B switch
	if (!no_deep && d.patch_grid == 0 && d.K % BK == 0 && a.a_colsum == nullptr && d.a_colsum == nullptr && tiles * a.splits <= 256 && kt_per >= 3 &&
		(d.a_layout == VITK_K_MAJOR || d.M >= 8) && (d.b_layout == VITK_K_MAJOR || d.N >= 8))
		a.dma = 3;
	dim3 grid((unsigned)tiles, (unsigned)a.splits);
	const bool patch = d.patch_grid > 0;
	int rc;
	if (patch) rc = launch<VITK_K_MAJOR, VITK_K_MAJOR, true>(a, d.out_dtype, grid, st);
	else if (d.a_layout == VITK_K_MAJOR && d.b_layout == VITK_K_MAJOR) rc = launch<VITK_K_MAJOR, VITK_K_MAJOR, false>(a, d.out_dtype, grid, st);
	else if (d.a_layout == VITK_K_MAJOR && d.b_layout == VITK_MN_MAJOR) rc = launch<VITK_K_MAJOR, VITK_MN_MAJOR, false>(a, d.out_dtype, grid, st);
	else if (d.a_layout == VITK_MN_MAJOR && d.b_layout == VITK_MN_MAJOR) rc = launch<VITK_MN_MAJOR, VITK_MN_MAJOR, false>(a, d.out_dtype, grid, st);
	else rc = launch<VITK_MN_MAJOR, VITK_K_MAJOR, false>(a, d.out_dtype, grid, st);
	if (rc == VITK_OK && a.partials != nullptr) {
		const ReduceItem r{a.partials, reinterpret_cast<float*>(d.C), d.M, d.N, d.ldc, a.tiles_n, (int)tiles, a.splits, d.accumulate, a.cs_partials, d.a_colsum, a.splits * a.cs_turns};
		hipLaunchKernelGGL(split_reduce128_kernel, dim3(BM * BN / 256, (unsigned)tiles + 1), dim3(256), 0, st, r);
		VITK_CHECK_LAUNCH();
	}
	if (rc == VITK_OK && fuse_colsum && colsum_done != nullptr) *colsum_done = true;
	return rc;
}

// A problem that vitk_gemm would run on the 128 x 128 kernel with K cut through the scratch buffer.
bool vitk_gemm_bf16_group_ok(const vitk_gemm_desc& d) {
	static const bool off = getenv("VITK_NO_GEMM_GROUP") != nullptr;  // A-B switch
	if (off || d.in_dtype != VITK_BF16 || !vitk_gemm_bf16_ok(d)) return false;
	if (vitk_gemm_bf16_256_ok(d)) {
		// ... or on 256 x 256 tiles, but with so few tiles and so short a K that its launch would leave more than half of the CUs
		// idle (384 x 384 ... 1536 x 384 over 4096 tokens: 4 ... 12 tiles x 8 slices)
		static const bool starved_off = getenv("VITK_NO_GROUP_STARVED") != nullptr;  // A-B switch
		const int64_t t256 = vitk_cdiv(d.M, 256) * vitk_cdiv(d.N, 256), kt = d.K / BK;
		const int64_t s256 = std::max<int64_t>(1, std::min<int64_t>(256 / t256, kt / 8));  // as vitk_gemm_bf16_256 cuts K
		if (starved_off || t256 * s256 > 128) return false;
	}
	if (d.a_layout != VITK_MN_MAJOR || d.b_layout != VITK_MN_MAJOR || d.patch_grid > 0 || !splits_k_128(d)) return false;
	return d.workspace != nullptr && vitk_aligned(d.workspace, 16);
}

int vitk_gemm_group_max() { return GROUP_MAX; }

// n <= GROUP_MAX independent problems, each vitk_gemm_bf16_group_ok and all with the same scratch buffer, as one launch plus
// one reduction launch.  VITK_ERR_UNSUPPORTED (nothing launched): the scratch buffer is too small.
int vitk_gemm_bf16_group128(const vitk_gemm_desc* const* ds, int n, hipStream_t st) {
	if (n < 1 || n > GROUP_MAX) return VITK_ERR_UNSUPPORTED;
	BfGroupArgs ga{};
	bool fused[GROUP_MAX];
	int64_t tiles[GROUP_MAX], ktiles[GROUP_MAX], work = 0;
	for (int i = 0; i < n; ++i) {
		fused[i] = fill128(*ds[i], ga.p[i]);
		if (ds[i]->a_colsum != nullptr && !fused[i]) return VITK_ERR_UNSUPPORTED;
		tiles[i] = (int64_t)ga.p[i].tiles_m * ga.p[i].tiles_n;
		ktiles[i] = vitk_cdiv(ds[i]->K, BK);
		work += tiles[i] * ktiles[i];
	}
	// one K-slice length for all problems (every workgroup then has the same amount of work), the shortest with which the
	// launch fits the chip at two workgroups per CU and no problem has more than 64 slices
	// (many tiles: the slices grow until every problem is one slice, and the launch simply has more rounds)
	int64_t kt_per = std::max<int64_t>(8, vitk_cdiv(work, 512)), longest = 0, ws_bytes = ds[0]->workspace_bytes;
	for (int i = 0; i < n; ++i) {
		longest = std::max(longest, ktiles[i]);
		ws_bytes = std::min(ws_bytes, ds[i]->workspace_bytes);
	}
	for (; kt_per < longest; ++kt_per) {
		int64_t wgs = 0, most = 0;
		for (int i = 0; i < n; ++i) {
			const int64_t s = vitk_cdiv(ktiles[i], kt_per);
			wgs += tiles[i] * s;
			most = std::max(most, s);
		}
		if (wgs <= 512 && most <= 64) break;
	}
	ReduceGroup rg{};
	int64_t used = 0;  // bytes of the scratch buffer
	int wg0 = 0, row0 = 0;
	for (int i = 0; i < n; ++i) {
		const vitk_gemm_desc& d = *ds[i];
		BfArgs& a = ga.p[i];
		a.splits = (int)vitk_cdiv(ktiles[i], kt_per);
		a.k_per_split = (int)(kt_per * BK);
		ga.first[i] = wg0;
		wg0 += (int)(tiles[i] * a.splits);
		if (a.splits == 1) continue;  // (stores its result itself)
		a.partials = reinterpret_cast<float*>(reinterpret_cast<char*>(d.workspace) + used);
		used += (int64_t)a.splits * tiles[i] * BM * BN * (int64_t)sizeof(float);
		if (fused[i]) {
			a.cs_partials = reinterpret_cast<float*>(reinterpret_cast<char*>(d.workspace) + used);
			a.cs_turns = a.dma == 2 ? a.tiles_n : 1;
			used += vitk_cdiv((int64_t)a.splits * a.cs_turns * d.M * (int64_t)sizeof(float), 256) * 256;
		}
		rg.p[rg.n] = ReduceItem{a.partials, reinterpret_cast<float*>(d.C), d.M, d.N, d.ldc, a.tiles_n, (int)tiles[i], a.splits, d.accumulate, a.cs_partials, d.a_colsum, a.splits * a.cs_turns};
		rg.first[rg.n] = row0;
		row0 += (int)tiles[i] + 1;
		++rg.n;
	}
	ga.first[n] = wg0;
	ga.n = n;
	rg.first[rg.n] = row0;
	if (used > ws_bytes) return VITK_ERR_UNSUPPORTED;
	vitk_last_gemm_kernel = "gemm_bf16_group_kernel";
	static bool attr_set = false;
	if (!attr_set) {
		hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr_set = true;
	}
	hipLaunchKernelGGL(gemm_bf16_group_kernel, dim3((unsigned)wg0), dim3(256), LDS_BYTES, st, ga);
	VITK_CHECK_LAUNCH();
	if (rg.n > 0) {
		hipLaunchKernelGGL(split_reduce128_group_kernel, dim3(BM * BN / 256, (unsigned)row0), dim3(256), 0, st, rg);
		VITK_CHECK_LAUNCH();
	}
	return VITK_OK;
}
