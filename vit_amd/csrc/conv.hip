// CvT kernels: depthwise convolution on token-major activations, im2col / col2im for the conv token
// embedding, BatchNorm over token-major activations.
#include "vitk_common.h"

extern "C" int vitk_dwconv_fwd(const void*, const void*, void*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, void*) { return VITK_ERR_UNSUPPORTED; }
extern "C" int vitk_dwconv_bwd(const void*, const void*, const void*, void*, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, void*) { return VITK_ERR_UNSUPPORTED; }
extern "C" int vitk_im2col(const void*, void*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, void*) { return VITK_ERR_UNSUPPORTED; }
extern "C" int vitk_col2im(const void*, void*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int, void*) { return VITK_ERR_UNSUPPORTED; }
extern "C" int vitk_batchnorm_fwd(const void*, const float*, const float*, float*, float*, void*, float*, float*, int64_t, int64_t, float, float, int, int, void*) { return VITK_ERR_UNSUPPORTED; }
extern "C" int vitk_batchnorm_bwd(const void*, const void*, const float*, const float*, const float*, void*, float*, float*, int64_t, int64_t, int, int, void*) { return VITK_ERR_UNSUPPORTED; }
