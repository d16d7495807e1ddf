// Diagnostic harness: runs the persistent 256x256 GEMM on a ViT-B/16 shape with in-kernel timestamps and prints where a
// workgroup's time goes per tile.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DVITK_GEMM_STAMPS -I include
// -I vit_amd/csrc tools/diag/gemm_stamps.hip vit_amd/csrc/status.cpp -o diag_bin/gemm_stamps
#include "../../vit_amd/csrc/gemm_bf16_256.hip"

#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
	const int64_t M = 50432, N = argc > 1 ? atoi(argv[1]) : 2304, K = argc > 2 ? atoi(argv[2]) : 768;
	const int epi = argc > 3 ? atoi(argv[3]) : 0;
	g_stagger = argc > 4 ? atoi(argv[4]) : 0;
	const int b_mn = argc > 5 ? atoi(argv[5]) : 0;
	bf16_t *A, *B, *C, *AUX, *OUT2;
	float* bias;
	hipMalloc(&A, M * K * 2);
	hipMalloc(&B, N * K * 2);
	hipMalloc(&C, M * N * 2);
	hipMalloc(&AUX, M * N * 2);
	hipMalloc(&OUT2, M * N * 2);
	hipMalloc(&bias, N * 4);
	hipMemset(A, 0x3c, M * K * 2);
	hipMemset(B, 0x3c, N * K * 2);
	hipMemset(AUX, 0x3c, M * N * 2);
	hipMemset(bias, 0, N * 4);
	const int nblk = 4096, rounds = 16;
	hipMalloc(&g_stamps, nblk * rounds * 8 * 8);
	hipMemset(g_stamps, 0, nblk * rounds * 8 * 8);
	vitk_gemm_desc d{};
	d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N;
	d.in_dtype = d.out_dtype = VITK_BF16;
	if (b_mn) { d.b_layout = VITK_MN_MAJOR; d.ldb = N; }
	d.epilogue = epi;
	const int variant = argc > 6 ? atoi(argv[6]) : 0;  // 1: no pre-activation output, 2: identity activation
	if (epi == VITK_EPI_ACT) { d.act = variant == 2 ? VITK_ACT_NONE : VITK_ACT_GELU; d.bias = bias; d.out2 = variant == 1 ? nullptr : OUT2; d.ld_out2 = N; }
	if (epi == VITK_EPI_RESIDUAL) { d.bias = bias; d.aux = AUX; d.ld_aux = N; }
	if (epi == VITK_EPI_ACTGRAD) { d.act = VITK_ACT_GELU; d.aux = AUX; d.ld_aux = N; }
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int it = 0; it < 3; ++it) {
		hipEventRecord(e0, 0);
		bool cs = false;
		int rc = vitk_gemm_bf16_256(d, 0, &cs);
		hipEventRecord(e1, 0);
		hipDeviceSynchronize();
		float ms;
		hipEventElapsedTime(&ms, e0, e1);
		printf("rc=%d %.1f us  %.1f TFLOP/s\n", rc, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
	}
	std::vector<long long> h(nblk * rounds * 8);
	hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost);
	if (getenv("PP_SECTIONS")) {
		// ping-pong kernel: shader-clock stamps of one K step of workgroup 0, per wave, at every barrier
		for (int w = 0; w < 8; ++w) {
			const long long* q = &h[w * 16];
			printf("wave %d:", w);
			for (int i = 1; i < 5; ++i) printf(" %5lld", q[i] - q[i - 1]);
			printf("   | K step %lld cycles, start offset %lld\n", q[4] - q[0], q[0] - h[0]);
		}
		return 0;
	}
	long long tmin = -1, tmax = 0;
	for (int b = 0; b < nblk; ++b)
		for (int r = 0; r < rounds; ++r) {
			const long long* s = &h[(b * rounds + r) * 8];
			if (s[0] == 0) continue;
			if (tmin < 0 || s[0] < tmin) tmin = s[0];
			if (s[4] > tmax) tmax = s[4];
		}
	printf("kernel span by stamps: %.2f us\n", (tmax - tmin) * 0.01);
	if (getenv("VITK_NO_PERSISTENT")) {
		// one workgroup per tile: show the first 12 workgroups of CU-slot order and the mean
		double w[4] = {0, 0, 0, 0};
		int cnt = 0;
		for (int b = 0; b < nblk; ++b) {
			const long long* s = &h[(b * rounds) * 8];
			if (s[0] == 0) continue;
			++cnt;
			for (int i = 0; i < 4; ++i) w[i] += (s[i + 1] - s[i]) * 0.01;
			if (b % 300 == 0) printf("wg %4d start %7.2f | first stage %5.2f | main %6.2f | epi->LDS %5.2f | stores %5.2f\n", b, (s[0] - tmin) * 0.01, (s[1] - s[0]) * 0.01, (s[2] - s[1]) * 0.01, (s[3] - s[2]) * 0.01, (s[4] - s[3]) * 0.01);
		}
		printf("mean of %d: first stage %5.2f | main %6.2f | epi->LDS %5.2f | stores %5.2f\n", cnt, w[0] / cnt, w[1] / cnt, w[2] / cnt, w[3] / cnt);
		return 0;
	}
	for (int r = 0; r < rounds; ++r) {
		double w[4] = {0, 0, 0, 0}, start = 0;
		int cnt = 0;
		for (int b = 0; b < nblk; ++b) {
			const long long* s = &h[(b * rounds + r) * 8];
			if (s[0] == 0) continue;
			++cnt;
			start += (s[0] - tmin) * 0.01;
			w[1] += (s[2] - s[0]) * 0.01;
			w[2] += (s[3] - s[2]) * 0.01;
			w[3] += (s[4] - s[3]) * 0.01;
		}
		if (cnt == 0) break;
		printf("round %2d: %3d wgs  start %7.2f us | wait first stage %5.2f | main loop %6.2f | next-stage issue %5.2f | epilogue %5.2f\n", r, cnt, start / cnt, w[0] / cnt,
			w[1] / cnt, w[2] / cnt, w[3] / cnt);
	}
	return 0;
}
