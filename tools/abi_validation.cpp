// Walks the host-side validation paths of the vitk C ABI with malformed arguments: every call must return a negative
// status (before or instead of a kernel launch) and must not touch memory it was not given.  Built against the
// AddressSanitizer build of the library (make -C vit_amd/csrc asan-run); needs no GPU.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "vitk.h"

static int failures = 0;
#define EXPECT_NEG(call)                                                          \
	do {                                                                          \
		const int rc_ = (call);                                                   \
		if (rc_ >= 0) {                                                           \
			std::printf("FAIL %s returned %d\n", #call, rc_);                     \
			++failures;                                                           \
		} else if (vitk_strerror(rc_) == nullptr || !*vitk_strerror(rc_)) {       \
			std::printf("FAIL %s: status %d has no text\n", #call, rc_);          \
			++failures;                                                           \
		}                                                                         \
	} while (0)

int main() {
	if (vitk_version() < 100) {
		std::printf("FAIL version %d\n", vitk_version());
		++failures;
	}
	for (int code = 0; code >= -8; --code)
		if (vitk_strerror(code) == nullptr) ++failures;
	if (vitk_strerror(VITK_ERR_HIP - 1) == nullptr || vitk_strerror(-123456) == nullptr) ++failures;

	alignas(256) static char buf[4096];
	void* p = buf;  // host memory standing in for device pointers: validation must reject before anything is launched on it
	float* f = reinterpret_cast<float*>(buf);

	vitk_gemm_desc d;
	std::memset(&d, 0, sizeof d);
	EXPECT_NEG(vitk_gemm(nullptr, nullptr));
	EXPECT_NEG(vitk_gemm(&d, nullptr));  // null operands
	d.A = d.B = p;
	d.C = p;
	EXPECT_NEG(vitk_gemm(&d, nullptr));  // zero sizes
	d.M = d.N = d.K = 64;
	d.lda = d.ldb = 8;  // leading dimensions smaller than the rows
	d.ldc = 64;
	EXPECT_NEG(vitk_gemm(&d, nullptr));
	d.lda = d.ldb = d.ldc = 64;
	d.in_dtype = 7;
	EXPECT_NEG(vitk_gemm(&d, nullptr));  // unknown dtype
	d.in_dtype = VITK_BF16;
	d.out_dtype = VITK_BF16;
	d.accumulate = 1;
	EXPECT_NEG(vitk_gemm(&d, nullptr));  // accumulate needs float32 output
	d.accumulate = 0;
	d.epilogue = 9;
	EXPECT_NEG(vitk_gemm(&d, nullptr));
	d.epilogue = VITK_EPI_NONE;
	d.drop_p[0] = 0.5f;
	EXPECT_NEG(vitk_gemm(&d, nullptr));  // dropout without an activation / residual epilogue
	d.drop_p[0] = 0.f;
	{  // several products in one call: every descriptor is checked before anything is launched
		vitk_gemm_desc two[2];
		two[0] = d;
		two[1] = d;
		two[0].in_dtype = two[0].out_dtype = two[1].in_dtype = two[1].out_dtype = VITK_F32;
		two[1].ldc = 1;  // the SECOND one is malformed
		EXPECT_NEG(vitk_gemm_group(two, 2, nullptr));
		EXPECT_NEG(vitk_gemm_group(nullptr, 2, nullptr));
		EXPECT_NEG(vitk_gemm_group(two, -1, nullptr));
		if (vitk_gemm_group(nullptr, 0, nullptr) != 0) {
			std::printf("FAIL vitk_gemm_group of nothing\n");
			++failures;
		}
	}
	d.epilogue = VITK_EPI_RESIDUAL;
	EXPECT_NEG(vitk_gemm(&d, nullptr));  // residual epilogue without aux
	d.epilogue = VITK_EPI_NONE;
	d.M = -1;
	EXPECT_NEG(vitk_gemm(&d, nullptr));
	d.M = int64_t(1) << 40;
	d.N = int64_t(1) << 40;
	EXPECT_NEG(vitk_gemm(&d, nullptr));

	EXPECT_NEG(vitk_layernorm_fwd(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 64, 1e-5f, VITK_F32, nullptr));
	EXPECT_NEG(vitk_layernorm_fwd(p, nullptr, f, f, p, nullptr, f, f, 0, 64, 1e-5f, VITK_F32, nullptr));
	EXPECT_NEG(vitk_layernorm_fwd(p, nullptr, f, f, p, nullptr, f, f, 4, 64, 1e-5f, 5, nullptr));
	EXPECT_NEG(vitk_attention_fwd(nullptr, p, p, p, f, 1, 1, 16, 16, 64, 64, 64, 64, 64, 0.f, VITK_BF16, nullptr));
	EXPECT_NEG(vitk_attention_fwd(p, p, p, p, f, 0, 1, 16, 16, 64, 64, 64, 64, 64, 0.f, VITK_BF16, nullptr));
	EXPECT_NEG(vitk_attention_fwd(p, p, p, p, f, 1, 1, 16, 16, 128, 128, 128, 128, 128, 0.f, VITK_BF16, nullptr));  // head size > 64
	EXPECT_NEG(vitk_attention_fwd(p, p, p, p, f, 1, 2, 16, 16, 64, 64, 64, 64, 64, 0.f, VITK_BF16, nullptr));      // row stride < heads * head size
	EXPECT_NEG(vitk_attention_fwd(p, p, p, p, f, int64_t(1) << 33, 1, 16, 16, 64, 64, 64, 64, 64, 0.f, VITK_BF16, nullptr));
	EXPECT_NEG(vitk_attention_fwd_dropout(p, p, p, p, f, 1, 1, 16, 16, 64, 64, 64, 64, 64, 0.f, 1.5f, 0, VITK_BF16, nullptr));
	EXPECT_NEG(vitk_attention_bwd(p, p, p, p, p, f, p, p, p, 1, 1, 16, 16, 64, 64, 64, 64, 64, 64, 64, 64, 0.f, nullptr, VITK_BF16, nullptr));  // no scratch
	const int64_t big = (int64_t(1) << 31) - 1;
	EXPECT_NEG(vitk_attention_bwd_noalloc(p, p, p, p, p, f, p, p, p, big, big, big, 16, 64, 64, 64, 64, 64, 64, 64, 64, 0.f, 0.f, 0, VITK_BF16, nullptr));
	EXPECT_NEG(vitk_attention_bwd_noalloc(p, p, p, p, p, f, p, p, p, 1024, 12, 197, 197, 64, 768, 768, 768, 768, 768, 768, 768, 0.f, 0.f, 0, VITK_BF16, nullptr));
	EXPECT_NEG(vitk_attention_bwd_noalloc(p, p, p, p, p, f, p, p, p, 0, 12, 197, 197, 64, 768, 768, 768, 768, 768, 768, 768, 0.f, 0.f, 0, VITK_BF16, nullptr));
	EXPECT_NEG(vitk_cast(nullptr, p, 16, VITK_F32, VITK_BF16, nullptr));
	EXPECT_NEG(vitk_cast(p, p, 16, 3, VITK_BF16, nullptr));
	EXPECT_NEG(vitk_dropout(p, p, 16, 1.0f, 0, VITK_F32, nullptr));
	EXPECT_NEG(vitk_dual_head_fwd(p, p, f, p, f, p, p, 4, 1, 64, 10, VITK_BF16, nullptr));  // fewer than two token rows
	EXPECT_NEG(vitk_ce_kd_loss(p, nullptr, nullptr, f, p, 4, 10, 0.f, 0.f, 1.f, VITK_F32, nullptr));
	EXPECT_NEG(vitk_ce_kd_loss(p, reinterpret_cast<const int64_t*>(buf), nullptr, f, p, 4, 10, 0.f, 0.f, 0.f, VITK_F32, nullptr));  // tau = 0
	EXPECT_NEG(vitk_batchnorm_act_fwd(nullptr, f, f, f, f, p, f, f, 64, 8, 1e-5f, 0.1f, 1, VITK_ACT_GELU, VITK_BF16, nullptr, 0, nullptr));
	EXPECT_NEG(vitk_batchnorm_act_fwd(p, f, f, f, f, p, f, f, 64, 8, 1e-5f, 0.1f, 1, 9, VITK_BF16, nullptr, 0, nullptr));               // unknown activation
	EXPECT_NEG(vitk_batchnorm_act_fwd(p, f, f, f, f, p, f, f, 64, 12, 1e-5f, 0.1f, 1, VITK_ACT_GELU, VITK_BF16, nullptr, 0, nullptr));  // channels not in 16-byte groups
	EXPECT_NEG(vitk_batchnorm_act_bwd(p, p, f, nullptr, f, f, p, f, f, 64, 8, 1, VITK_ACT_SILU, VITK_BF16, nullptr, 0, nullptr));       // the activation needs beta
	EXPECT_NEG(vitk_dwconv_bwd_acc(p, p, 8, p, p, 8, nullptr, 1, 1, 4, 4, 8, 3, 1, 1, VITK_BF16, nullptr));
	EXPECT_NEG(vitk_dwconv_bwd_acc(p, p, 4, p, p, 8, f, 1, 1, 4, 4, 8, 3, 1, 1, VITK_BF16, nullptr));  // row stride below the channel count
	std::printf(failures == 0 ? "abi validation: all calls rejected as expected\n" : "abi validation: %d failures\n", failures);
	return failures == 0 ? 0 : 1;
}
