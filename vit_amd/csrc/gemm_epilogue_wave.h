// Epilogue of one wave's 128 x 64 accumulator tile (8 x 4 tiles of v_mfma_f32_16x16x32_bf16 with the weight fragment as
// the first operand: acc[i][j][r] is row 16 i + (lane & 15), column 16 j + 4 (lane >> 4) + r) through a wave-private
// 4 KiB LDS image, shared by the bf16-output GEMM kernels.
#pragma once

#include <type_traits>

#include "gemm_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
	if constexpr (I < N) {
		f(std::integral_constant<int, I>{});
		static_for<I + 1, N>(f);
	}
}

// Per-tile epilogue operands a lane keeps in registers: bias / LayerScale of its 8 columns.  They are fetched by inline-asm
// loads (NOT waited for here): a load the compiler can see would make it wait for ALL outstanding memory operations --
// the operand pipeline of the main loop included -- wherever a register it thinks the load may still be writing gets
// reused.  epilogue_wave retires them with its first wait.
struct EpiRegs {
	f32x4 bias[2], gamma[2];
};

template <int EPI>
__device__ __forceinline__ void epilogue_prefetch(const GemmDev& g, EpiRegs& er, bool full, int64_t n0w, int lane) {
	const uint32_t voff = 32u * (uint32_t)(lane & 7);
	er.bias[0] = er.bias[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
	er.gamma[0] = er.gamma[1] = (f32x4){1.f, 1.f, 1.f, 1.f};
	if (!full) return;
	if (g.bias != nullptr) {
		const float* p = g.bias + n0w;  // wave-uniform
		asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(er.bias[0]) : "v"(voff), "s"(p) : "memory");
		asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(er.bias[1]) : "v"(voff), "s"(p) : "memory");
	}
	if (EPI == VITK_EPI_RESIDUAL && g.gamma != nullptr) {
		const float* p = g.gamma + n0w;
		asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(er.gamma[0]) : "v"(voff), "s"(p) : "memory");
		asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(er.gamma[1]) : "v"(voff), "s"(p) : "memory");
	}
}

// Epilogue of one wave's 128 x 64 accumulator tile, 16 rows (one row of MFMA tiles) at a time: the wave writes the
// four accumulator fragments (accumulator layout: 16 rows x 16 bytes per wave-instruction) into its private image,
// reads them back as rows (each lane 8 consecutive columns of one row) and does the fused arithmetic there, so that
// operands arrive and results leave in 16-byte accesses, 8 rows x 128 contiguous bytes per wave-instruction.  LDS
// operations of one wave complete in order, so no barrier is needed.  `full`: wave-uniform, the wave's 128 x 64 block
// lies inside M x N and every operand may be accessed 16 bytes wide.
template <int EPI, int ACT, bool EDGE = true>
__device__ __forceinline__ void epilogue_wave(const GemmDev& g, const f32x4 (&acc)[8][4], EpiRegs& er, bool full, char* img, int64_t m0, int64_t n0w, int lane) {
	const int c8 = lane & 7, r8 = lane >> 3;
	const int64_t nb = n0w + 8 * c8;
	const bf16_t* aux = reinterpret_cast<const bf16_t*>(g.aux);
	bf16_t* out2 = reinterpret_cast<bf16_t*>(g.out2);
	bf16_t* outp = reinterpret_cast<bf16_t*>(g.C);
	constexpr bool HAS_AUX = EPI == VITK_EPI_RESIDUAL || EPI == VITK_EPI_ACTGRAD;
	const int wrow = lane & 15;
	auto write_rows = [&](auto ic) {
		constexpr int i = decltype(ic)::value;
#pragma unroll
		for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(img + wrow * 256 + (((4 * j + (lane >> 4)) ^ wrow) << 4)) = acc[i][j];
	};
	if (EDGE && !full) {
		static_for<0, 8>([&](auto ic) {
			constexpr int i = decltype(ic)::value;
			write_rows(ic);
#pragma unroll 1
			for (int idx = lane; idx < 16 * 64; idx += 64) {
				const int row = idx >> 6, col = idx & 63;
				const int64_t m = m0 + 16 * i + row, n = n0w + col;
				const float v = *reinterpret_cast<const float*>(img + row * 256 + ((((col >> 2) ^ row) << 4) | ((col & 3) << 2)));
				if (m < g.M && n < g.N) epi_store1<bf16_t, bf16_t, EPI, ACT>(g, m, n, v);
			}
		});
		return;
	}
	const bool drop_on = g.drop[0].thr16 != 0;
	// Operand rows (residual / pre-activation) of the NEXT 16 rows are requested one step ahead as inline asm (a
	// wave-uniform row base in scalar registers plus one 32-bit lane offset) and retired by a COUNTED wait that leaves the
	// newer requests and the stores in flight; written as plain loads the compiler sinks them to their uses and waits
	// vmcnt(0) there, i.e. for every store issued before them.  The "memory" clobbers keep the stores between the asm
	// statements, which the counts rely on.
	u32x4 auxv[2][2];
	const uint32_t aux_voff = (uint32_t)(r8 * (int)g.ld_aux + 8 * c8) * 2u;
	constexpr int NST = (EPI == VITK_EPI_ACT) ? 4 : 2;  // stores per 16 rows when out2 is written (the count below assumes the maximum is fine: see wait)
	auto request = [&](int i, u32x4(&dst)[2]) {
#pragma unroll
		for (int it = 0; it < 2; ++it) {
			const bf16_t* rowbase = aux + (m0 + 16 * i + 8 * it) * g.ld_aux + n0w;  // wave-uniform
			asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[it]) : "v"(aux_voff), "s"(rowbase) : "memory");
		}
	};
	if (HAS_AUX) request(0, auxv[0]);
	else asm volatile("s_waitcnt vmcnt(0)" : "+v"(er.bias[0]), "+v"(er.bias[1])::"memory");  // the bias loads (and whatever was in flight before them)
	static_for<0, 8>([&](auto ic) {
		constexpr int i = decltype(ic)::value;
		if (HAS_AUX) {
			if constexpr (i < 7) request(i + 1, auxv[(i + 1) & 1]);
			u32x4(&cur)[2] = auxv[i & 1];
			// in flight behind the rows needed now: the 2 stores of the previous step (i > 0) and the 2 requests just issued (i < 7)
			if constexpr (i == 0)
				asm volatile("s_waitcnt vmcnt(2)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(er.bias[0]), "+v"(er.bias[1]), "+v"(er.gamma[0]), "+v"(er.gamma[1])::"memory");
			else if constexpr (i < 7) asm volatile("s_waitcnt vmcnt(4)" : "+v"(cur[0]), "+v"(cur[1])::"memory");
			else asm volatile("s_waitcnt vmcnt(2)" : "+v"(cur[0]), "+v"(cur[1])::"memory");
		}
		write_rows(ic);
#pragma unroll
		for (int it = 0; it < 2; ++it) {
			const int row = 8 * it + r8;
			const int64_t m = m0 + 16 * i + row;
			const f32x4 lo = *reinterpret_cast<const f32x4*>(img + row * 256 + (((2 * c8) ^ row) << 4));
			const f32x4 hi = *reinterpret_cast<const f32x4*>(img + row * 256 + (((2 * c8 + 1) ^ row) << 4));
			float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
			for (int e = 0; e < 8; ++e) v[e] += er.bias[e >> 2][e & 3];
			float keep[8];
			if (EPI != VITK_EPI_NONE && drop_on) {  // fused dropout (wave-uniform branch)
				drop_keep_n<8>(g, m, nb, keep);
				if (EPI != VITK_EPI_ACT) {
#pragma unroll
					for (int e = 0; e < 8; ++e) v[e] *= keep[e];
				}
			}
			if (EPI == VITK_EPI_ACT) {
				if (out2 != nullptr) {
					bf16x8 pre;
#pragma unroll
					for (int e = 0; e < 8; ++e) {
						pre[e] = (bf16_t)v[e];
						v[e] = (float)pre[e];  // the activation sees the pre-activation as stored
					}
					*reinterpret_cast<bf16x8*>(out2 + m * g.ld_out2 + nb) = pre;
				}
				if constexpr (ACT == VITK_ACT_GELU) {
					gelu_bf16_n<8>(v);
				} else {
#pragma unroll
					for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e], g.act);
				}
				if (drop_on) {
#pragma unroll
					for (int e = 0; e < 8; ++e) v[e] *= keep[e];
				}
			} else if (EPI == VITK_EPI_RESIDUAL) {
#pragma unroll
				for (int e = 0; e < 8; ++e) v[e] = (float)__builtin_bit_cast(bf16x8, auxv[i & 1][it])[e] + er.gamma[e >> 2][e & 3] * v[e];
			} else if (EPI == VITK_EPI_ACTGRAD) {
				float u[8];
#pragma unroll
				for (int e = 0; e < 8; ++e) u[e] = (float)__builtin_bit_cast(bf16x8, auxv[i & 1][it])[e];
				if constexpr (ACT == VITK_ACT_GELU) {
					gelu_deriv_bf16_n<8>(v, u);
				} else {
#pragma unroll
					for (int e = 0; e < 8; ++e) v[e] *= act_deriv(u[e], g.act);
				}
			}
			bf16x8 pk;
#pragma unroll
			for (int e = 0; e < 8; ++e) pk[e] = (bf16_t)v[e];
			*reinterpret_cast<bf16x8*>(outp + m * g.ldc + nb) = pk;
		}
	});
	(void)NST;
}


}  // namespace
