// C ABI of the attention core: argument validation and dispatch between the bf16 MFMA kernels (whole key set in LDS for
// head size 64 and at most 256 tokens; blocked online softmax for the rest) and the generic vector-ALU kernels.
#include "attention.h"

#include <mutex>

static int check_common(const AttnArgs& a) {
	VITK_REQUIRE(a.B > 0 && a.n_head > 0 && a.Sq > 0 && a.Skv > 0 && a.dh > 0, VITK_ERR_SHAPE);
	VITK_REQUIRE(a.dh <= 64, VITK_ERR_SHAPE);
	VITK_REQUIRE(a.B <= 65535 && a.n_head <= 65535, VITK_ERR_SHAPE);
	const int64_t hd = (int64_t)a.n_head * a.dh;
	VITK_REQUIRE(a.ldq >= hd && a.ldk >= hd && a.ldv >= hd && a.ldo >= hd, VITK_ERR_SHAPE);
	return VITK_OK;
}

// dropout_p in [0, 1): quantised to 1/65536 (the kept probabilities are rescaled by the quantised keep rate, so the
// expectation of the output is exact); 0 switches dropout off
static int set_dropout(AttnArgs& a, float dropout_p, uint64_t seed) {
	VITK_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, VITK_ERR_SHAPE);
	uint32_t thr = (uint32_t)lrintf(dropout_p * 65536.f);
	if (thr > 65535u) thr = 65535u;
	if (dropout_p > 0.f && thr == 0) thr = 1;
	a.drop_thr16 = thr;
	a.drop_scale = 65536.f / (float)(65536u - thr);
	a.seed_lo = (uint32_t)seed;
	a.seed_hi = (uint32_t)(seed >> 32);
	a.drop_epoch = thr != 0 ? vitk_dropout_epoch_current() : nullptr;
	return VITK_OK;
}

extern "C" int vitk_attention_fwd_dropout(const void* q, const void* k, const void* v, void* out, float* lse, int64_t B, int64_t n_head, int64_t Sq,
	int64_t Skv, int64_t dh, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, float dropout_p, uint64_t seed, int dtype, void* stream) {
	VITK_REQUIRE(q && k && v && out && lse, VITK_ERR_NULL);
	VITK_REQUIRE(B < (1ll << 31) && n_head < (1ll << 31) && Sq < (1ll << 31) && Skv < (1ll << 31) && dh < (1ll << 31), VITK_ERR_SHAPE);
	AttnArgs a{};
	a.q = q; a.k = k; a.v = v; a.out = out; a.out_w = out; a.lse = lse;
	a.B = (int)B; a.n_head = (int)n_head; a.Sq = (int)Sq; a.Skv = (int)Skv; a.dh = (int)dh;
	a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
	a.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)dh);
	int rc = check_common(a);
	if (rc != VITK_OK) return rc;
	rc = set_dropout(a, dropout_p, seed);
	if (rc != VITK_OK) return rc;
	if (vitk_attention_mfma_ok(a, dtype)) return vitk_attention_mfma(a, false, (hipStream_t)stream);
	if (vitk_attention_flash_ok(a, dtype)) return vitk_attention_flash(a, false, (hipStream_t)stream);
	return vitk_attention_generic(a, false, dtype, (hipStream_t)stream);
}

extern "C" int vitk_attention_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int64_t B, int64_t n_head, int64_t Sq,
	int64_t Skv, int64_t dh, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, int dtype, void* stream) {
	return vitk_attention_fwd_dropout(q, k, v, out, lse, B, n_head, Sq, Skv, dh, ldq, ldk, ldv, ldo, scale, 0.f, 0, dtype, stream);
}

extern "C" int vitk_attention_bwd_dropout(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse, void* dq,
	void* dk, void* dv, int64_t B, int64_t n_head, int64_t Sq, int64_t Skv, int64_t dh, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
	int64_t lddq, int64_t lddk, int64_t lddv, float scale, float dropout_p, uint64_t seed, float* scratch, int dtype, void* stream) {
	VITK_REQUIRE(dout && q && k && v && out && lse && dq && dk && dv && scratch, VITK_ERR_NULL);
	VITK_REQUIRE(B < (1ll << 31) && n_head < (1ll << 31) && Sq < (1ll << 31) && Skv < (1ll << 31) && dh < (1ll << 31), VITK_ERR_SHAPE);
	AttnArgs a{};
	a.q = q; a.k = k; a.v = v; a.out = out; a.dout = dout; a.lse = const_cast<float*>(lse);
	a.dq = dq; a.dk = dk; a.dv = dv;
	a.bwd_scratch = scratch;
	a.B = (int)B; a.n_head = (int)n_head; a.Sq = (int)Sq; a.Skv = (int)Skv; a.dh = (int)dh;
	a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
	a.scale = scale > 0.f ? scale : 1.0f / sqrtf((float)dh);
	int rc = check_common(a);
	if (rc != VITK_OK) return rc;
	const int64_t hd = (int64_t)a.n_head * a.dh;
	VITK_REQUIRE(lddq >= hd && lddk >= hd && lddv >= hd, VITK_ERR_SHAPE);
	rc = set_dropout(a, dropout_p, seed);
	if (rc != VITK_OK) return rc;
	if (vitk_attention_mfma_ok(a, dtype)) return vitk_attention_mfma(a, true, (hipStream_t)stream);
	if (vitk_attention_flash_ok(a, dtype)) return vitk_attention_flash(a, true, (hipStream_t)stream);
	return vitk_attention_generic(a, true, dtype, (hipStream_t)stream);
}

extern "C" int vitk_attention_bwd(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse, void* dq,
	void* dk, void* dv, int64_t B, int64_t n_head, int64_t Sq, int64_t Skv, int64_t dh, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
	int64_t lddq, int64_t lddk, int64_t lddv, float scale, float* scratch, int dtype, void* stream) {
	return vitk_attention_bwd_dropout(dout, q, k, v, out, lse, dq, dk, dv, B, n_head, Sq, Skv, dh, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, 0.f, 0, scratch,
		dtype, stream);
}

// ---- the backward for callers that cannot allocate ----------------------------------------------------------------
// Same computation as vitk_attention_bwd_dropout with the softmax-statistics scratch (2 * B * n_head * Sq floats) taken
// from a fixed array in the library's device image instead of a caller buffer.  Every GPU the library runs on holds its
// own copy of the array at its own address until the process ends; the address is looked up once per device, with that
// device current, on the first call there.  All calls on one GPU share the array: calls that overlap (different streams)
// race.  Nothing in the array has to start at zero (the dQ pass writes every word the dK/dV pass reads).
constexpr int64_t STATIC_SCRATCH_FLOATS = 2ll << 20;  // 8 MiB
static __device__ __align__(256) float scratch[STATIC_SCRATCH_FLOATS];

static int static_scratch_ptr(float** out) {
	constexpr int MAXDEV = 64;
	static float* ptr[MAXDEV] = {nullptr};
	static std::mutex mu;
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
	if (dev < 0 || dev >= MAXDEV) return VITK_ERR_UNSUPPORTED;
	std::lock_guard<std::mutex> lock(mu);
	if (ptr[dev] == nullptr) {
		float* p = nullptr;
		e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));  // `scratch` itself is only a host-side handle, never a device address
		if (e != hipSuccess) return VITK_ERR_HIP - (int)e;
		if (p == nullptr) return VITK_ERR_NULL;
		ptr[dev] = p;
	}
	*out = ptr[dev];
	return VITK_OK;
}

extern "C" int vitk_attention_bwd_noalloc(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse, void* dq,
	void* dk, void* dv, int64_t B, int64_t n_head, int64_t Sq, int64_t Skv, int64_t dh, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
	int64_t lddq, int64_t lddk, int64_t lddv, float scale, float dropout_p, uint64_t seed, int dtype, void* stream) {
	VITK_REQUIRE(B > 0 && n_head > 0 && Sq > 0, VITK_ERR_SHAPE);
	VITK_REQUIRE(B < (1ll << 31) && n_head < (1ll << 31) && Sq < (1ll << 31), VITK_ERR_SHAPE);
	// 2 * B * n_head * Sq <= STATIC_SCRATCH_FLOATS, compared factor by factor (the product of three values below 2^31 does
	// not fit in 64 bits)
	constexpr int64_t CAP = STATIC_SCRATCH_FLOATS / 2;
	if (B > CAP || n_head > CAP / B || Sq > CAP / (B * n_head)) return VITK_ERR_SCRATCH;
	float* p = nullptr;
	const int rc = static_scratch_ptr(&p);
	if (rc != VITK_OK) return rc;
	return vitk_attention_bwd_dropout(dout, q, k, v, out, lse, dq, dk, dv, B, n_head, Sq, Skv, dh, ldq, ldk, ldv, ldo, lddq, lddk, lddv, scale, dropout_p, seed, p,
		dtype, stream);
}
