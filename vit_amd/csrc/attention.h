// Argument block shared by the attention kernels.
#pragma once

#include "vitk_common.h"

struct AttnArgs {
	const void *q, *k, *v, *out, *dout;
	float* lse;
	void *dq, *dk, *dv;
	void* out_w;  // forward output (same pointer as `out` in the backward, where it is read-only)
	int B, n_head, Sq, Skv, dh;
	int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv;
	float scale;
	// attention-probability dropout (F.scaled_dot_product_attention's dropout_p, layers/attention.py:41,86): a
	// probability is dropped when its 16 mask bits are below drop_thr16 (= p * 65536, 0 = no dropout) and the kept
	// ones are scaled by drop_scale = 65536 / (65536 - drop_thr16)
	uint32_t drop_thr16, seed_lo, seed_hi;
	float drop_scale;
	const uint32_t* drop_epoch;  // device word mixed into the seed (see DropKey::epoch), or null
	float* bwd_scratch;  // backward: [2][B * n_head * Sq] floats (lse in log2 units, delta = rowsum(dO o O)): dQ kernel -> dK/dV kernel
	int heads_per_wg;  // set by the launcher: consecutive (batch, head) pairs one workgroup of the forward kernel walks
	uint32_t dq_owner;  // one-pass backward, set by its launcher: wave (3 bits each) that forms dQ tile t = 4 * (16-query block) + (16-column block) of a step
};

// The dropout mask is a pure function of (seed, batch * n_head + head, query, key), so the forward and both backward
// kernels regenerate it instead of storing it: one 32-bit hash per (query, pair of keys), 16 bits per key.
__host__ __device__ __forceinline__ uint32_t vitk_hash32(uint32_t x) {
	x ^= x >> 16;
	x *= 0x7feb352du;
	x ^= x >> 15;
	x *= 0x846ca68bu;
	x ^= x >> 16;
	return x;
}
// once per kernel, on the kernel's own argument block, before the first mask (see drop_key_enter)
__device__ __forceinline__ void attn_enter(AttnArgs& a) {
	if (a.drop_thr16 != 0u && a.drop_epoch != nullptr) a.seed_lo += drop_epoch_term(a.drop_epoch);
}
__device__ __forceinline__ uint32_t attn_head_key(const AttnArgs& a, int bh) { return vitk_hash32(a.seed_lo ^ vitk_hash32(a.seed_hi ^ (uint32_t)bh)); }
__device__ __forceinline__ uint32_t attn_row_key(uint32_t head_key, int q) { return vitk_hash32(head_key + (uint32_t)q * 0x9E3779B1u); }
__device__ __forceinline__ uint32_t attn_pair_bits(uint32_t row_key, int key_pair) { return vitk_hash32(row_key ^ ((uint32_t)key_pair * 0x85EBCA6Bu + 0x1B873593u)); }
// multiplier (0 or drop_scale) of the probability of key `key`, given the bits of its pair
__device__ __forceinline__ float attn_keep(uint32_t pair_bits, int key, uint32_t thr16, float drop_scale) {
	const uint32_t bits = (key & 1) ? (pair_bits >> 16) : (pair_bits & 0xffffu);
	return bits >= thr16 ? drop_scale : 0.f;
}

int vitk_attention_generic(const AttnArgs& a, bool bwd, int dtype, hipStream_t st);
// bf16 MFMA kernels; return VITK_ERR_UNSUPPORTED when the shape does not qualify
bool vitk_attention_mfma_ok(const AttnArgs& a, int dtype);
int vitk_attention_mfma(const AttnArgs& a, bool bwd, hipStream_t st);
// the backward of the above in one kernel (S and dP formed once): at most 224 keys, more than 32 queries; VITK_ATTN_BWD_FUSED=0
// keeps the two-kernel form
bool vitk_attention_bwd_fused_ok(const AttnArgs& a);
int vitk_attention_bwd_fused(const AttnArgs& a, hipStream_t st);
// bf16 MFMA kernels for any sequence length and head sizes 4, 8, ..., 64 (keys walked in blocks of 64, online softmax); no dropout
bool vitk_attention_flash_ok(const AttnArgs& a, int dtype);
int vitk_attention_flash(const AttnArgs& a, bool bwd, hipStream_t st);
