// Stand-alone A/B bench of GEMM kernels on random bf16 data: the library's vitk_gemm against vitk_gemm_bf16_t128,
// interleaved in one process, outputs compared element by element.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ivit_amd/csrc tools/diag/gemm_lab.hip vit_amd/csrc/gemm_bf16_t128.hip -Lvit_amd -lvitk -o diag_bin/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_common.h"

int vitk_gemm_bf16_t128(const vitk_gemm_desc& d, hipStream_t st);
int vitk_gemm_bf16_p8(const vitk_gemm_desc& d, hipStream_t st);
int vitk_t128_occupancy(int lds);
#ifdef VITK_T128_STAMPS
void vitk_t128_set_stamps(long long* p);
#endif
#ifdef VITK_P8_STAMPS
void vitk_p8_set_stamps(long long* p);
#endif

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(bf16_t* p, int64_t n, uint32_t seed, float scale) {
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		uint32_t x = (uint32_t)i * 2654435761u + seed;
		x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
		p[i] = (bf16_t)(((float)(x & 0xffff) / 32768.f - 1.f) * scale);
	}
}
__global__ void fill_f32(float* p, int64_t n, uint32_t seed) {
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		uint32_t x = (uint32_t)i * 2654435761u + seed;
		x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15;
		p[i] = (float)(x & 0xffff) / 32768.f - 1.f;
	}
}
__global__ void diff_kernel(const uint16_t* a, const uint16_t* b, int64_t n, unsigned long long* nd) {
	unsigned long long c = 0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) c += a[i] != b[i];
	if (c) atomicAdd(nd, c);
}

int main(int argc, char** argv) {
	struct Shape { const char* name; int64_t M, N, K; };
	std::vector<Shape> shapes = {{"qkv", 50432, 2304, 768}, {"proj", 50432, 768, 768}, {"fc_in", 50432, 3072, 768}, {"fc_out", 50432, 768, 3072},
		{"dqkv", 50432, 768, 2304}, {"sq4k", 4096, 4096, 4096}, {"sq8k", 8192, 8192, 8192}, {"rag", 5000, 1000 - 8, 256}};
	const int iters = argc > 1 ? atoi(argv[1]) : 10;
	const bool use_p8 = getenv("LAB_KERNEL") && !strcmp(getenv("LAB_KERNEL"), "p8");
	auto cand = [&](const vitk_gemm_desc& d, hipStream_t s_) { return use_p8 ? vitk_gemm_bf16_p8(d, s_) : vitk_gemm_bf16_t128(d, s_); };
	const char* only = argc > 2 ? argv[2] : nullptr;
	hipStream_t st;
	CK(hipStreamCreate(&st));
	if (only && !strcmp(only, "occ")) {
		for (int lds = 65536; lds <= 81920; lds += 1024) printf("lds %d: %d workgroups per CU\n", lds, vitk_t128_occupancy(lds));
		return 0;
	}
	const int epis[4] = {VITK_EPI_NONE, VITK_EPI_ACT, VITK_EPI_RESIDUAL, VITK_EPI_ACTGRAD};
	const char* enames[4] = {"plain", "gelu+pre", "residual", "gelu'"};
	for (const Shape& s : shapes) {
		if (only && strcmp(only, s.name)) continue;
		bf16_t *A, *B, *C0, *C1, *aux, *p0, *p1;
		float* bias;
		unsigned long long* nd;
		CK(hipMalloc(&A, s.M * s.K * 2)); CK(hipMalloc(&B, s.N * s.K * 2));
		CK(hipMalloc(&C0, s.M * s.N * 2)); CK(hipMalloc(&C1, s.M * s.N * 2)); CK(hipMalloc(&aux, s.M * s.N * 2));
		CK(hipMalloc(&p0, s.M * s.N * 2)); CK(hipMalloc(&p1, s.M * s.N * 2));
		CK(hipMalloc(&bias, s.N * 4)); CK(hipMalloc(&nd, 8));
		fill_bf16<<<2048, 256, 0, st>>>(A, s.M * s.K, 1u, 1.f);
		fill_bf16<<<2048, 256, 0, st>>>(B, s.N * s.K, 2u, 0.05f);
		fill_bf16<<<2048, 256, 0, st>>>(aux, s.M * s.N, 3u, 2.f);
		fill_f32<<<64, 256, 0, st>>>(bias, s.N, 4u);
		for (int e = 0; e < 4; ++e) {
			vitk_gemm_desc d{};
			d.A = A; d.B = B; d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.K; d.ldb = s.K; d.ldc = s.N;
			d.a_layout = VITK_K_MAJOR; d.b_layout = VITK_K_MAJOR; d.in_dtype = VITK_BF16; d.out_dtype = VITK_BF16;
			d.epilogue = epis[e]; d.act = e == 1 || e == 3 ? VITK_ACT_GELU : VITK_ACT_NONE;
			d.bias = e == 3 ? nullptr : bias;
			if (e >= 2) { d.aux = aux; d.ld_aux = s.N; }
			vitk_gemm_desc d0 = d, d1 = d;
			d0.C = C0; d1.C = C1;
			if (e == 1) { d0.out2 = p0; d1.out2 = p1; d0.ld_out2 = d1.ld_out2 = s.N; }
			CK(hipMemsetAsync(C0, 0, s.M * s.N * 2, st)); CK(hipMemsetAsync(C1, 0xff, s.M * s.N * 2, st));
			int rc0 = vitk_gemm(&d0, st), rc1 = cand(d1, st);
			if (rc0 || rc1) { printf("%s %s: rc %d %d\n", s.name, enames[e], rc0, rc1); continue; }
			CK(hipMemsetAsync(nd, 0, 8, st));
			diff_kernel<<<1024, 256, 0, st>>>((const uint16_t*)C0, (const uint16_t*)C1, s.M * s.N, nd);
			if (e == 1) diff_kernel<<<1024, 256, 0, st>>>((const uint16_t*)p0, (const uint16_t*)p1, s.M * s.N, nd);
			unsigned long long h = 0;
			CK(hipMemcpyAsync(&h, nd, 8, hipMemcpyDeviceToHost, st));
			CK(hipStreamSynchronize(st));
			float best[2] = {1e9f, 1e9f};
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
			for (int rep = 0; rep < 3; ++rep)
				for (int w = 0; w < 2; ++w) {
					CK(hipEventRecord(e0, st));
					for (int i = 0; i < iters; ++i) { if (w == 0) vitk_gemm(&d0, st); else cand(d1, st); }
					CK(hipEventRecord(e1, st));
					CK(hipEventSynchronize(e1));
					float ms; CK(hipEventElapsedTime(&ms, e0, e1));
					best[w] = std::min(best[w], ms / iters);
				}
#ifdef VITK_T128_STAMPS
			{
				const int nb = 1024;
				long long* sd;
				CK(hipMalloc(&sd, nb * 16 * 4 * 8));
				CK(hipMemset(sd, 0, nb * 16 * 4 * 8));
				vitk_t128_set_stamps(sd);
				vitk_gemm_bf16_t128(d1, st);
				CK(hipStreamSynchronize(st));
				vitk_t128_set_stamps(nullptr);
				std::vector<long long> hs(nb * 16 * 4);
				CK(hipMemcpy(hs.data(), sd, nb * 16 * 4 * 8, hipMemcpyDeviceToHost));
				hipFree(sd);
				// per round: mean main loop, epilogue; kernel span; per-workgroup first start
				long long t0 = 1ll << 62, t1 = 0;
				for (int b = 0; b < nb; ++b) if (hs[(b * 16) * 4] ) { t0 = std::min(t0, hs[(b * 16) * 4]); }
				double ml[16] = {0}, ep[16] = {0}; int cnt[16] = {0};
				for (int b = 0; b < nb; ++b) for (int r = 0; r < 16; ++r) {
					const long long* q = &hs[(b * 16 + r) * 4];
					if (!q[0]) continue;
					ml[r] += q[1] - q[0]; ep[r] += q[2] - q[1]; cnt[r]++; t1 = std::max(t1, q[2]);
				}
				printf("   stamps (cycles of s_memtime):");
				for (int r = 0; r < 12 && cnt[r]; ++r) printf(" r%d[n=%d ml %.0f ep %.0f]", r, cnt[r], ml[r] / cnt[r], ep[r] / cnt[r]);
				printf("\n   workgroups 0,256 (same CU?) hwid %llx %llx xcc %lld %lld start %lld %lld; wg0 rounds:", hs[3] & 0xffffffff, hs[(256 * 16) * 4 + 3] & 0xffffffff, hs[3] >> 32, hs[(256 * 16) * 4 + 3] >> 32,
					hs[0] - t0, hs[(256 * 16) * 4] - t0);
				for (int r = 0; r < 6; ++r) printf(" [%lld %lld %lld]", hs[r * 4] - t0, hs[r * 4 + 1] - t0, hs[r * 4 + 2] - t0);
				printf("\n   wg256 rounds:");
				for (int r = 0; r < 6; ++r) printf(" [%lld %lld %lld]", hs[(256 * 16 + r) * 4] - t0, hs[(256 * 16 + r) * 4 + 1] - t0, hs[(256 * 16 + r) * 4 + 2] - t0);
				printf("\n");
			}
#endif
#ifdef VITK_P8_STAMPS
			if (use_p8) {
				const int nb = 1024;
				const size_t bytes = (nb * 16 * 4 + 8 * 16) * 8;
				long long* sd;
				CK(hipMalloc(&sd, bytes));
				CK(hipMemset(sd, 0, bytes));
				vitk_p8_set_stamps(sd);
				vitk_gemm_bf16_p8(d1, st);
				CK(hipStreamSynchronize(st));
				vitk_p8_set_stamps(nullptr);
				std::vector<long long> hs(bytes / 8);
				CK(hipMemcpy(hs.data(), sd, bytes, hipMemcpyDeviceToHost));
				hipFree(sd);
				double ml[16] = {0}, ep[16] = {0}; int cnt[16] = {0};
				for (int b = 0; b < nb; ++b) for (int r = 0; r < 16; ++r) {
					const long long* q = &hs[(b * 16 + r) * 4];
					if (!q[0]) continue;
					ml[r] += q[1] - q[0]; ep[r] += q[2] - q[1]; cnt[r]++;
				}
				printf("   stamps:");
				for (int r = 0; r < 12 && cnt[r]; ++r) printf(" r%d[n=%d ml %.0f ep %.0f]", r, cnt[r], ml[r] / cnt[r], ep[r] / cnt[r]);
				printf("\n");
				for (int w = 0; w < 8; w += 4) {
					const long long* q = &hs[nb * 16 * 4 + w * 16];
					printf("   wave %d K step 3, deltas between barriers:", w);
					for (int i = 1; i < 5; ++i) printf(" %lld", q[i] - q[i - 1]);
					printf("  (total %lld)\n", q[4] - q[0]);
				}
			}
#endif
			const double fl = 2.0 * s.M * s.N * s.K / 1e9;
			printf("%-7s %-9s M=%lld N=%lld K=%lld: old %8.1f us %7.1f TF | new %8.1f us %7.1f TF | mismatches %llu\n", s.name, enames[e], (long long)s.M, (long long)s.N,
				(long long)s.K, best[0] * 1e3, fl / best[0], best[1] * 1e3, fl / best[1], h);
			fflush(stdout);
		}
		hipFree(A); hipFree(B); hipFree(C0); hipFree(C1); hipFree(aux); hipFree(p0); hipFree(p1); hipFree(bias); hipFree(nd);
	}
	return 0;
}
