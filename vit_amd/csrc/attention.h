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
};

int vitk_attention_generic(const AttnArgs& a, bool bwd, int dtype, hipStream_t st);
// bf16 MFMA kernels; return VITK_ERR_UNSUPPORTED when the shape does not qualify
bool vitk_attention_mfma_ok(const AttnArgs& a, int dtype);
int vitk_attention_mfma(const AttnArgs& a, bool bwd, hipStream_t st);
