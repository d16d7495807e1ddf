// bf16 MFMA attention kernels (placeholder until the MFMA path is enabled).
#include "attention.h"

bool vitk_attention_mfma_ok(const AttnArgs&, int) { return false; }
int vitk_attention_mfma(const AttnArgs&, bool, hipStream_t) { return VITK_ERR_UNSUPPORTED; }
