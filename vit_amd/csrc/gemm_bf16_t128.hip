// bf16 MFMA GEMM for K-major x K-major operands and bf16 outputs (forward and input-gradient GEMMs of nn.Linear,
// layers/transformers.py:46-54 and layers/attention.py:28,47 of the reference), built for TWO workgroups per CU:
// 128 x 256 x 64 tile per 256-thread workgroup (4 waves side by side, 128 x 64 per wave = 8 x 4 tiles of
// v_mfma_f32_16x16x32_bf16, 128 accumulator registers), 80 KiB of LDS per workgroup.
//
// Why two workgroups per CU: a workgroup alternates between phases that use different parts of the CU -- the MFMA
// main loop, the wait for operands, and the epilogue (VALU arithmetic such as GELU, operand loads, output stores).
// With one workgroup per CU those phases run one after the other; with two independent workgroups the hardware
// interleaves them, so one workgroup's epilogue and operand waits hide behind the other's MFMAs, the stores of
// the chip no longer come in synchronised bursts, and the tail of a launch is balanced at half the granularity.
//
// LDS: [A stage 0: 16 KiB][A stage 1: 16 KiB][B: 32 KiB][epilogue image: 4 x 4 KiB].  A (128 rows x 64 k) is double
// buffered.  B (256 rows x 64 k) is SINGLE buffered: at the top of a K step every wave copies its 8 B fragments
// (its own 64 columns, both halves of the K step) into registers, and after a barrier the next K step's B tile is
// fetched into the same buffer while the 64 MFMAs of this step run.  Operands arrive by LDS-DMA
// (global_load_lds_dwordx4) with the bank-conflict swizzle applied to the per-lane SOURCE address (chunk ^= row & 7)
// and to the fragment reads.
//
// The K steps of consecutive tiles form one pipeline (the next tile's first K step is fetched during the last K step
// of the current one), and the epilogue runs per wave through a wave-private 16 x 64 float image with no barrier.
#include <cstdlib>
#include <type_traits>

#include "gemm_common.h"
#include "gemm_epilogue_wave.h"

namespace {

constexpr int TM = 128, TN = 256, BK = 64, NTHREADS = 256;
constexpr int A_BYTES = TM * BK * 2;   // 16 KiB
constexpr int B_BYTES = TN * BK * 2;   // 32 KiB
constexpr int B_OFF = 2 * A_BYTES;
constexpr int EPI_OFF = B_OFF + B_BYTES;
constexpr int EW_BYTES = 16 * 64 * 4;  // 4 KiB per wave: 16 rows x 64 columns of float, 16-byte chunk ^= row
constexpr int LDS_BYTES = EPI_OFF + 4 * EW_BYTES;  // 81,920 B: two workgroups fill the CU's 160 KiB exactly

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* gbl_void_p;

__device__ __forceinline__ int off_k(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

__device__ __forceinline__ bf16x8 read_frag(const char* tile, int base, int s, int lane) {
	return *reinterpret_cast<const bf16x8*>(tile + off_k(base + (lane & 15), 4 * s + (lane >> 4)));
}

__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(lds_void_p)(void*)p; }

// 16-byte LDS read issued as inline asm (NOT waited for: the caller retires it with an s_waitcnt that names `dst`)
__device__ __forceinline__ void lds_read128(bf16x8& dst, uint32_t addr, int offset) {
	asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(offset));
}

// LDS-DMA of 8 rows x 128 bytes (one KiB) of a K-major operand tile; rows past the end re-read the last row (their
// products are never stored).  The source address is a wave-uniform base plus a 32-bit lane offset recomputed on every
// call (the empty asm keeps the compiler from carrying loop-invariant address registers through the MFMA loop).
__device__ __forceinline__ void stage_piece(char* tile, const bf16_t* __restrict__ base, int64_t ld, int64_t mn0, int64_t mn_total, int k0, int piece, int lane) {
	asm volatile("" : "+v"(lane));
	const int ldi = (int)ld;
	const int r = lane >> 3;  // == row & 7
	const int chunk = (lane & 7) ^ r;
	int64_t rowbase = mn0 + 8 * piece;
	if (rowbase > mn_total - 1) rowbase = mn_total - 1;
	const int rlim = (int)(mn_total - 1 - rowbase);
	const int rr = r < rlim ? r : rlim;
	const uint32_t voff = (uint32_t)(rr * ldi + 8 * chunk) * 2u;
	const char* src = reinterpret_cast<const char*>(base + rowbase * ld + k0) + voff;
	__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(tile + piece * 1024), 16, 0, 0);
}

struct T128Args {
	GemmDev g;
	int tiles_m, tiles_n;
	int vec8;          // epilogue operands / outputs may be accessed 16 bytes wide
	int stagger;       // start delay of the second half of the grid, in units of 1024 cycles
#ifdef VITK_T128_STAMPS
	long long* stamps;  // diagnostic builds only: [workgroup][round < 16][4] = {tile start, main loop end, epilogue end, XCC id << 8 | CU id}
#endif
	unsigned* counter; // dynamic tile scheduling: 8 counters (one per XCD, 64 bytes apart), zeroed before the launch; or null
};

template <int ACT>
__device__ __forceinline__ float act_fwd1(float u, int act_rt) {
	if constexpr (ACT == VITK_ACT_GELU) return gelu_bf16(u);
	else return act_apply(u, act_rt);
}

#ifdef VITK_T128_STAMPS
long long* g_t128_stamps = nullptr;
#define STAMP(slot) \
	if (a.stamps != nullptr && tid == 0 && round < 16) a.stamps[((int64_t)blockIdx.x * 16 + round) * 4 + (slot)] = __builtin_readcyclecounter()
#else
#define STAMP(slot)
#endif

template <int EPI, int ACT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_t128_kernel(T128Args a) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	gemm_dev_enter(a.g);
	const GemmDev& g = a.g;
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wn = wave * 64;
	const int ntiles = a.tiles_m * a.tiles_n;
	const int nblk = gridDim.x;  // a multiple of 8
	const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
	constexpr int GM = 8;  // row tiles per raster group: the 64 workgroups that share an L2 then work on an 8 x 8 block of tiles
	const int per_group = GM * a.tiles_n;

	// tile of this workgroup in a round: consecutive tiles (in grouped order) go to workgroups that share an XCD
	auto tile_of = [&](int round, int64_t& m0, int64_t& n0) -> bool {
		const int base = round * nblk;
		const int n = min(nblk, ntiles - base);
		if (n <= 0) return false;  // (m0, n0 untouched)
		const int q = n >> 3, r = n & 7;
		if (idx >= q + (xcd < r ? 1 : 0)) return false;
		const int t = base + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
		const int group = t / per_group;
		const int first_m = group * GM;
		const int gsz = min(a.tiles_m - first_m, GM);
		const int tm = first_m + (t % per_group) % gsz;
		const int tn = (t % per_group) / gsz;
		m0 = (int64_t)tm * TM;
		n0 = (int64_t)tn * TN;
		return true;
	};

	const int nt = (int)(g.K / BK);
	const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
	const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
	char* const sB = smem + B_OFF;
	// fragment read addresses: row (lane & 15) of a 16-row block, 16-byte chunk (lane >> 4) ^ (row & 7) of the first half
	// of the K step; the second half is the same address with bit 6 flipped (chunk + 4), the next 16 rows are +2048
	const uint32_t lds0 = lds_addr(smem);
	const uint32_t a_lane = (uint32_t)off_k(lane & 15, lane >> 4);
	const uint32_t b_lane = (uint32_t)off_k(wn + (lane & 15), lane >> 4);
	char* const img = smem + EPI_OFF + wave * EW_BYTES;

	// LDS-DMA source offsets of a lane inside an 8-row piece (row lane >> 3, 16-byte chunk (lane & 7) ^ row): loop invariant
	const uint32_t a_voff = (uint32_t)((lane >> 3) * (int)g.lda + 8 * ((lane & 7) ^ (lane >> 3))) * 2u;
	const uint32_t b_voff = (uint32_t)((lane >> 3) * (int)g.ldb + 8 * ((lane & 7) ^ (lane >> 3))) * 2u;
	// piece j of this wave's share of a tile that lies inside the matrix: a scalar base plus the lane offset
	auto stage_a = [&](int buf, int64_t sm0, int k0, int j) {
		const char* src = reinterpret_cast<const char*>(A + (sm0 + 32 * wave + 8 * j) * g.lda + k0) + a_voff;
		__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(smem + buf * A_BYTES + (4 * wave + j) * 1024), 16, 0, 0);
	};
	auto stage_b = [&](int64_t sn0, int k0, int j) {
		const char* src = reinterpret_cast<const char*>(B + (sn0 + 64 * wave + 8 * j) * g.ldb + k0) + b_voff;
		__builtin_amdgcn_global_load_lds((gbl_void_p)src, (lds_void_p)(sB + (8 * wave + j) * 1024), 16, 0, 0);
	};
	auto stage = [&](int buf, int64_t sm0, int64_t sn0, int k0) {
		if (sm0 + TM <= g.M && sn0 + TN <= g.N) {
#pragma unroll
			for (int j = 0; j < 4; ++j) stage_a(buf, sm0, k0, j);
#pragma unroll
			for (int j = 0; j < 8; ++j) stage_b(sn0, k0, j);
		} else {
#pragma unroll
			for (int j = 0; j < 4; ++j) stage_piece(smem + buf * A_BYTES, A, g.lda, sm0, g.M, k0, 4 * wave + j, lane);
#pragma unroll
			for (int j = 0; j < 8; ++j) stage_piece(sB, B, g.ldb, sn0, g.N, k0, 8 * wave + j, lane);
		}
	};

	int64_t m0 = 0, n0 = 0;
	bool have = tile_of(0, m0, n0);
	int par = 0;  // A buffer of the current K step
	// Two workgroups that share a CU overlap best when one is in its epilogue (VALU, stores) while the other is in its
	// main loop (MFMA); started together they stay in step.  The second half of the grid (by dispatch order the second
	// workgroup of each CU; a guess that only affects speed) therefore starts about half a tile late.
	if (blockIdx.x >= (gridDim.x >> 1))
		for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
	if (have) stage(0, m0, n0, 0);
	for (int round = 0; have; ++round) {
		STAMP(0);
		f32x4 acc[8][4];
#pragma unroll
		for (int i = 0; i < 8; ++i)
#pragma unroll
			for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
		int64_t nm0 = m0, nn0 = n0;  // after the last tile: re-fetch this tile's first K step (nobody reads it)
		const bool nxt = tile_of(round + 1, nm0, nn0);
		for (int t = 0; t < nt; ++t) {
			// this K step's operands have landed (every wave waits for its own requests, then the barrier)
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_s_barrier();
			// Fragment reads as inline asm: for plain LDS loads the compiler may put an s_waitcnt vmcnt(0) in front of a read
			// that follows an LDS-DMA request (it cannot tell that the request targets the OTHER buffer), which would make
			// this K step wait for the next one's operands.  Each wait names the registers it retires, so no MFMA that uses
			// them can be scheduled above it.
			const uint32_t sa = lds0 + par * A_BYTES + a_lane;
			bf16x8 wf[2][4];
#pragma unroll
			for (int s = 0; s < 2; ++s)
#pragma unroll
				for (int j = 0; j < 4; ++j) lds_read128(wf[s][j], (lds0 + B_OFF + b_lane) ^ (s << 6), 2048 * j);
			bf16x8 xf[3];
			lds_read128(xf[0], sa, 0);
			lds_read128(xf[1], sa, 2048);
			asm volatile("s_waitcnt lgkmcnt(2)"
						 : "+v"(wf[0][0]), "+v"(wf[0][1]), "+v"(wf[0][2]), "+v"(wf[0][3]), "+v"(wf[1][0]), "+v"(wf[1][1]), "+v"(wf[1][2]), "+v"(wf[1][3])
						 :
						 : "memory");
			__builtin_amdgcn_s_barrier();  // every wave holds its B fragments: the B buffer may be overwritten
			int t1 = t + 1;
			asm volatile("" : "+s"(t1));  // opaque: keeps the compiler from peeling the last K step into a second copy of the loop body
			const bool last = t1 == nt;
			stage(par ^ 1, last ? nm0 : m0, last ? nn0 : n0, last ? 0 : (t + 1) * BK);
			static_for<0, 16>([&](auto gc) {
				constexpr int gi = decltype(gc)::value, s = gi >> 3, i = gi & 7;
				if constexpr (gi + 2 < 16) lds_read128(xf[(gi + 2) % 3], sa ^ (((gi + 2) >> 3) << 6), 2048 * ((gi + 2) & 7));
				if constexpr (gi + 2 < 16) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(xf[gi % 3])::"memory");
				else if constexpr (gi + 1 < 16) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(xf[gi % 3])::"memory");
				else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[gi % 3])::"memory");
#pragma unroll
				for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], xf[gi % 3], acc[i][j], 0, 0, 0);
			});
			par ^= 1;
		}
		STAMP(1);
		// (keeps the epilogue's operand loads out of the last K step, where a wait for them would also wait for the DMA)
		asm volatile("" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
		const bool full = a.vec8 && n0 + wn + 64 <= g.N && m0 + TM <= g.M;
		EpiRegs er;
		epilogue_prefetch<EPI>(g, er, full, n0 + wn, lane);
		epilogue_wave<EPI, ACT>(g, acc, er, full, img, m0, n0 + wn, lane);
		STAMP(2);
#ifdef VITK_T128_STAMPS
		if (a.stamps != nullptr && tid == 0 && round < 16) {
			unsigned hwid, xcc;
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
			a.stamps[((int64_t)blockIdx.x * 16 + round) * 4 + 3] = ((long long)(xcc & 15) << 32) | hwid;
		}
#endif
		m0 = nm0;
		n0 = nn0;
		have = nxt;
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last (unused) DMA must not outlive the workgroup's LDS
}

int t128_blocks() {
	static int n = 0;
	if (n == 0) {
		int dev = 0, cus = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
		n = 2 * (cus & ~7);
		if (const char* e = std::getenv("VITK_T128_BLOCKS")) n = atoi(e);
	}
	return n;
}

template <int EPI, int ACT>
int launch_t128(const T128Args& a, hipStream_t st) {
	static bool attr = false;
	if (!attr) {
		hipError_t e = hipFuncSetAttribute((const void*)gemm_bf16_t128_kernel<EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		attr = true;
	}
	const int64_t tiles = (int64_t)a.tiles_m * a.tiles_n;
	int blocks = t128_blocks();
	if (tiles < blocks) blocks = (int)((tiles + 7) & ~7ll);
	hipLaunchKernelGGL((gemm_bf16_t128_kernel<EPI, ACT>), dim3((unsigned)blocks), dim3(NTHREADS), LDS_BYTES, st, a);
	VITK_CHECK_LAUNCH();
	return VITK_OK;
}

}  // namespace

#ifdef VITK_T128_STAMPS
void vitk_t128_set_stamps(long long* p) { g_t128_stamps = p; }
#endif

// diagnostics: resident workgroups per CU the runtime reports for the plain variant at `lds` bytes of dynamic LDS
int vitk_t128_occupancy(int lds) {
	int n = -1;
	if (hipFuncSetAttribute((const void*)gemm_bf16_t128_kernel<VITK_EPI_NONE, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_bf16_t128_kernel<VITK_EPI_NONE, -1>, NTHREADS, lds) != hipSuccess) return -3;
	return n;
}

bool vitk_gemm_bf16_t128_ok(const vitk_gemm_desc& d) {
	if (d.patch_grid > 0 || d.tokens_per_image > 0 || d.K % BK != 0 || d.K < BK) return false;
	if (d.a_layout != VITK_K_MAJOR || d.b_layout != VITK_K_MAJOR) return false;
	if (d.in_dtype != VITK_BF16 || d.out_dtype != VITK_BF16 || d.accumulate) return false;
	if (d.M < 128 || d.N < 64) return false;
	return true;
}

int vitk_gemm_bf16_t128(const vitk_gemm_desc& d, hipStream_t st) {
	vitk_last_gemm_kernel = "gemm_bf16_t128_kernel";
	T128Args a{};
	a.g = make_gemm_dev(d);
	a.tiles_m = (int)vitk_cdiv(d.M, TM);
	a.tiles_n = (int)vitk_cdiv(d.N, TN);
	auto al16 = [](const void* p) { return p == nullptr || vitk_aligned(p, 16); };
	{
		static const char* e = std::getenv("VITK_T128_STAGGER");
		a.stagger = e ? atoi(e) : 0;
	}
#ifdef VITK_T128_STAMPS
	a.stamps = g_t128_stamps;
#endif
	a.vec8 = (d.N % 8 == 0) && (d.ldc % 8 == 0) && (d.ld_aux % 8 == 0) && (d.ld_out2 % 8 == 0) && al16(d.C) && al16(d.bias) && al16(d.gamma) && al16(d.aux) && al16(d.out2);
	switch (d.epilogue) {
		case VITK_EPI_NONE: return launch_t128<VITK_EPI_NONE, -1>(a, st);
		case VITK_EPI_ACT: return d.act == VITK_ACT_GELU ? launch_t128<VITK_EPI_ACT, VITK_ACT_GELU>(a, st) : launch_t128<VITK_EPI_ACT, -1>(a, st);
		case VITK_EPI_RESIDUAL: return launch_t128<VITK_EPI_RESIDUAL, -1>(a, st);
		case VITK_EPI_ACTGRAD: return d.act == VITK_ACT_GELU ? launch_t128<VITK_EPI_ACTGRAD, VITK_ACT_GELU>(a, st) : launch_t128<VITK_EPI_ACTGRAD, -1>(a, st);
		default: return VITK_ERR_UNSUPPORTED;
	}
}
