/*
 * vitk — C ABI of the MI355X (gfx950) kernel library behind the ViT/CvT hot path.
 *
 * The reference (newbee1905/vit) has no FFI: its hot path is a chain of PyTorch ATen
 * calls inside nn.Module.forward bodies.  Each entry point below replaces the ATen
 * call(s) of one reference call site (cited as file:line under the reference
 * checkout) and is what the drop-in layer modules (vit_amd/layers) bind through ctypes.
 *
 * Conventions
 *  - every buffer is caller-allocated DEVICE memory; nothing is allocated or freed here;
 *  - every call is asynchronous on the `stream` argument (a hipStream_t passed as void*)
 *    and never synchronises; distinct streams may be used from distinct threads;
 *  - `dtype` arguments: 0 = float32, 1 = bfloat16.  Parameters of norms, biases,
 *    LayerScale vectors and statistics (mean/rstd/lse) are always float32;
 *  - return value: 0 on success, a negative vitk_status otherwise (HIP runtime errors are
 *    reported as VITK_ERR_HIP - hipError_t); vitk_strerror() gives the text.  Shapes and
 *    alignments are validated on the host before any kernel is launched.
 */
#ifndef VITK_H
#define VITK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITK_VERSION 100

enum vitk_status {
	VITK_OK = 0,
	VITK_ERR_SHAPE = -1,      /* a dimension is non-positive or exceeds what the kernel supports */
	VITK_ERR_DTYPE = -2,      /* dtype code not supported by this entry point */
	VITK_ERR_ALIGN = -3,      /* pointer / leading dimension not aligned as the kernel requires */
	VITK_ERR_NULL = -4,       /* a required pointer is null */
	VITK_ERR_UNSUPPORTED = -5,/* combination of options not implemented */
	VITK_ERR_SCRATCH = -6,    /* the problem exceeds the fixed scratch of a *_noalloc entry point */
	VITK_ERR_HIP = -1000      /* -1000 - hipError_t */
};

enum vitk_dtype { VITK_F32 = 0, VITK_BF16 = 1 };

/* activation codes (layers/transformers.py:26-33; F.gelu is the exact erf form) */
enum vitk_act { VITK_ACT_NONE = 0, VITK_ACT_GELU = 1, VITK_ACT_SILU = 2, VITK_ACT_RELU = 3, VITK_ACT_RELU6 = 4 };

/* operand layouts of vitk_gemm: which index is contiguous in memory */
enum vitk_layout {
	VITK_K_MAJOR = 0,   /* A(m,k) = A[m*lda + k]      B(n,k) = B[n*ldb + k] */
	VITK_MN_MAJOR = 1   /* A(m,k) = A[k*lda + m]      B(n,k) = B[k*ldb + n] */
};

/* epilogues of vitk_gemm; acc = sum_k A(m,k) B(n,k), v = acc + bias[n] (bias optional) */
enum vitk_epilogue {
	VITK_EPI_NONE = 0,      /* C = v                                                        */
	VITK_EPI_ACT = 1,       /* C = act(v);  out2 (optional) = v  (the pre-activation)        */
	VITK_EPI_RESIDUAL = 2,  /* C = aux + gamma[n] * v   (gamma optional = 1)                 */
	VITK_EPI_ACTGRAD = 3    /* C = v * act'(aux)        (aux = saved pre-activation)         */
};

const char* vitk_strerror(int status);
int vitk_version(void);

/* ------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[M,N] (+)= epilogue( sum_k A(m,k) * B(n,k) ).
 * Replaces every nn.Linear on the path and its autograd mm pair:
 *   forward  y = x W^T + b   models/vit.py:30,39,47,65; layers/attention.py:17-18,28,47,57-64,68-70,92;
 *                            layers/transformers.py:36-40,46,54          (A = x K-major, B = W K-major)
 *   dgrad    dx = dy W                                                   (A = dy K-major, B = W MN-major)
 *   wgrad    dW = dy^T x                                                 (A = dy MN-major, B = x MN-major)
 * and fuses bias, activation (layers/transformers.py:48-52), LayerScale + residual add
 * (layers/vit.py:44-47,52-55) and the activation derivative of the backward into the epilogue.
 *
 * Patch embedding (models/vit.py:46-52, layers/embeddings.py:7-16) is the same GEMM with an
 * im2col-free A operand: with patch_grid > 0, A is an NCHW image batch [*, img_c, img_hw, img_hw]
 * and A(m,k) = image[m / P2][c][py*p + i][px*p + j], P2 = patch_grid^2, p = img_hw / patch_grid,
 * (py,px) = divmod(m % P2, patch_grid), (c,i,j) = unravel(k, (img_c,p,p)).  With tokens_per_image > 0
 * output row m is stored at row m + (m / tokens_per_image + 1) * L and pos[(L + m % tokens_per_image) * N + n] is added to
 * v, where L = max(lead_rows, 1) rows are left free in front of every image: the class-token row of models/vit.py:49, and
 * with lead_rows = 2 a second learned (distillation) token row of a DeiT-style student.
 *
 * in_dtype is the dtype of A, B and aux; out_dtype of C and out2.  bias, gamma and pos are float32.
 * The bf16 MFMA path needs K % 8 == 0, 16-byte aligned A/B and leading dimensions that are
 * multiples of 8 elements; other bf16 shapes and all float32 inputs run on the generic
 * (float32-MFMA) path.  accumulate != 0 adds into C (out_dtype must be float32).
 */
typedef struct vitk_gemm_desc {
	const void* A;
	const void* B;
	void* C;
	int64_t M, N, K;
	int64_t lda, ldb, ldc;
	int32_t a_layout, b_layout;
	int32_t in_dtype, out_dtype;
	int32_t epilogue, act;
	int32_t accumulate;
	int32_t patch_grid;        /* 0 = A is a plain matrix */
	int32_t img_c, img_hw;
	int32_t tokens_per_image;  /* 0 = no row remap / no pos add */
	int32_t lead_rows;         /* free token rows in front of every image when tokens_per_image > 0; 0 and 1 both mean one */
	const float* bias;         /* [N] or null */
	const float* gamma;        /* [N] or null (VITK_EPI_RESIDUAL) */
	const float* pos;          /* [(tokens_per_image + max(lead_rows, 1)), N] or null */
	const void* aux;           /* [M, ld_aux] residual or pre-activation */
	int64_t ld_aux;
	void* out2;                /* [M, ld_out2] optional second output */
	int64_t ld_out2;
	float* a_colsum;           /* [M] or null; MN-major A only: a_colsum[m] += sum_k A(m,k), i.e. the bias gradient
	                              (column sums of dy) that autograd's addmm backward forms next to dW = dy^T x.
	                              ACCUMULATED into: the caller zeroes it. */
	void* workspace;           /* optional device scratch the call may use (it allocates nothing itself): with at
	                              least vitk_gemm_workspace_bytes() the K-split weight-gradient GEMM writes its partial
	                              tiles there and sums them in a second kernel (deterministic), instead of float
	                              atomics into C.  Contents are undefined afterwards; calls that share it must be
	                              ordered on one stream. */
	int64_t workspace_bytes;
	float drop_p[2];           /* dropout fused into the epilogue (0 = off), nn.Dropout of layers/vit.py:17-23 and
	                              layers/transformers.py:20-21: with VITK_EPI_ACT the activation output, with
	                              VITK_EPI_RESIDUAL the branch value (acc + bias) before gamma and the residual add,
	                              with VITK_EPI_ACTGRAD the incoming gradient (acc) are multiplied by the mask of
	                              element m * N + n and by 1 / keep-probability; two entries = two stacked dropouts.
	                              The mask is the one vitk_dropout(p, seed) applies to a contiguous [M, N] tensor.
	                              Any other epilogue with drop_p != 0: VITK_ERR_UNSUPPORTED. */
	uint64_t drop_seed[2];
	int64_t drop_row0;         /* the mask index of this problem's row 0 is drop_row0 * N: a caller that splits one
	                              [M, N] product into row blocks passes each block's first row, so that the blocks
	                              draw the mask of the whole tensor */
} vitk_gemm_desc;

int vitk_gemm(const vitk_gemm_desc* desc, void* stream);
/* n independent products in one call: the same results as vitk_gemm(&descs[i], stream) for i = 0 .. n - 1 in any order, so
 * no product may read or write another's C / a_colsum, and all descriptors are checked before anything is launched.  What it
 * is for: the four weight gradients of one block of a small model (dW = dy^T x of layers/attention.py:28,47 and
 * layers/transformers.py:46-54 at d_model 192: 4 ... 12 output tiles over tens of thousands of tokens) -- products with
 * MN-major operands, float32 result and the SAME workspace pointer share one launch, in which K is cut into a quarter of the
 * slices each would need alone (a quarter of the partial-tile traffic, one reduction launch instead of four).  Products the
 * grouped kernel does not take run exactly as vitk_gemm runs them.  The reference has no counterpart (autograd issues one
 * addmm per Linear). */
int vitk_gemm_group(const vitk_gemm_desc* descs, int n, void* stream);
/* Size of `workspace` that is enough for any problem on this device (a fixed number of 256x256 float32 tiles). */
int64_t vitk_gemm_workspace_bytes(void);
/* on != 0: the persistent GEMM kernels launch the FEWEST workgroups that finish a problem in the same number of rounds of
 * tiles (591 tiles: 197 workgroups x 3 instead of 256 of which 79 walk a third tile), so that the remaining CUs stay free
 * for kernels of other streams for the whole launch -- the communication kernels of a gradient all-reduce that overlaps the
 * backward pass (no counterpart in the reference, which has no distributed path).  Costs 0.2 ms of a 34 ms ViT-B/16 step
 * when nothing else runs, saves 4 % of it when another kernel holds 16 CUs half of the time.  Process-wide; returns the
 * previous setting.  Default off. */
int vitk_gemm_leave_cus_free(int on);
/* 1 if vitk_gemm would run this problem on the bf16 MFMA kernel, 0 if on the generic float32-MFMA kernel. */
int vitk_gemm_uses_bf16_mfma(const vitk_gemm_desc* desc);
/* Name of the kernel family the calling thread's last vitk_gemm launched ("gemm_bf16_p8_kernel", "gemm_bf16_p8w_kernel",
 * "gemm_bf16_t128_kernel", "gemm_bf16_256_kernel", "gemm_bf16_kernel", "gemm_bf16_deep_kernel", "gemm_bf16_group_kernel", "gemm_generic_kernel"; "" before the first call): the
 * names a kernel trace shows, for reports that put measured launch times next to a profile.  No counterpart in the
 * reference (torch.nn.functional.linear picks its kernel out of sight, layers/transformers.py:46-54). */
const char* vitk_gemm_last_kernel(void);

/* ------------------------------------------------------------------------------------------
 * Row normalisation.  rows = B*N tokens, d = d_model <= 4096 when d and the pointers allow 16-byte accesses
 * (d % 4 == 0 in float32, d % 8 == 0 in bfloat16), else d <= 1024; beyond that VITK_ERR_UNSUPPORTED.
 * LayerNorm: nn.LayerNorm injected through config.norm (layers/vit.py:15,20,38,49; eps 1e-5).
 * RMSNorm: layers/norm.py:19-31 (fp32 internal, y = x * rsqrt(mean(x^2)+eps) * (unit_offset + w)).
 * If `res` is non-null the kernel first forms s = x + res, writes s to `sum_out` (the new
 * residual stream, may alias x) and normalises s.
 * Backward: dx = d(loss)/d(x) of y = norm(x); if `dres` is non-null it is added to dx
 * (gradient flowing around the norm on the residual stream).  dgamma / dbeta are float32
 * [d] and are ACCUMULATED into (the caller zeroes them).  `workspace` (optional device scratch, 16-byte
 * aligned, at least 2048 * d bytes; the call allocates nothing itself): the per-workgroup column sums go
 * through it and a second small launch adds them up in a fixed order; without it they are added with float
 * atomics, whose same-address queue costs about 10 us per call.
 */
int vitk_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta,
	void* y, void* sum_out, float* mean, float* rstd, int64_t rows, int64_t d, float eps, int dtype, void* stream);
int vitk_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
	const void* dres, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t d, int dtype, void* workspace,
	int64_t workspace_bytes, void* stream);
int vitk_rmsnorm_fwd(const void* x, const void* res, const float* weight, void* y, void* sum_out, float* rstd,
	int64_t rows, int64_t d, float eps, int unit_offset, int dtype, void* stream);
int vitk_rmsnorm_bwd(const void* dy, const void* x, const float* weight, const float* rstd, const void* dres,
	void* dx, float* dweight, int64_t rows, int64_t d, int unit_offset, int dtype, void* workspace, int64_t workspace_bytes,
	void* stream);
/* DyT, the dynamic-tanh stand-in for a norm (layers/norm.py:35-45): y = gamma * tanh(alpha[0] * x) + beta over the last
 * dimension d (<= 4096 with 16-byte accessible rows and d % 4 == 0, else <= 1024).  Backward: dx, and dalpha (1 float),
 * dgamma, dbeta ([d] floats) ACCUMULATED into (the caller zeroes them). */
int vitk_dyt_fwd(const void* x, const float* alpha, const float* gamma, const float* beta, void* y, int64_t rows, int64_t d,
	int dtype, void* stream);
int vitk_dyt_bwd(const void* dy, const void* x, const float* alpha, const float* gamma, void* dx, float* dalpha, float* dgamma,
	float* dbeta, int64_t rows, int64_t d, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head attention core: F.scaled_dot_product_attention(q,k,v,is_causal=False)
 * (layers/attention.py:39-43,84-88) together with the head split/merge rearranges
 * (layers/attention.py:31-33,45,80-82,90).
 * q is [B, Sq, *] with row stride ldq; head h occupies columns [h*dh, (h+1)*dh) — i.e. the
 * 'b s (h d)' layout the projections produce, so q/k/v may be column slices of one fused
 * [B, S, 3*d_model] buffer.  k, v are [B, Skv, *].  out is [B, Sq, n_head*dh] with row stride ldo.
 * lse [B, n_head, Sq] float32 = log-sum-exp of the scaled scores, saved for the backward.
 * scale <= 0 selects 1/sqrt(dh).  Backward needs out and lse of the forward, and `scratch`: 2 * B * n_head * Sq
 * floats of caller-provided device memory (its dQ pass leaves the per-row softmax statistics there for its dK/dV pass).
 */
int vitk_attention_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
	int64_t B, int64_t n_head, int64_t Sq, int64_t Skv, int64_t dh,
	int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, int dtype, void* stream);
int vitk_attention_bwd(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse,
	void* dq, void* dk, void* dv,
	int64_t B, int64_t n_head, int64_t Sq, int64_t Skv, int64_t dh,
	int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
	float scale, float* scratch, int dtype, void* stream);
/* The bf16 backward for head size 64, at most 224 keys and more than 32 (at most 256) queries runs as ONE kernel that forms
 * the scores and their gradient once (five matrix products; `scratch` then only receives the stores of rows that do not
 * exist); other shapes, and every shape after vitk_attention_bwd_one_pass(0), run as a dQ kernel followed by a dK/dV kernel
 * that each recompute them (seven products).  Process-wide switch for comparisons (environment VITK_ATTN_BWD_FUSED=0 sets the
 * initial value); returns the previous setting.  No counterpart in the reference (autograd of layers/attention.py:39-43). */
int vitk_attention_bwd_one_pass(int on);
/* The same with dropout on the attention probabilities: the `dropout_p` argument of
 * F.scaled_dot_product_attention in training mode (layers/attention.py:41,86):
 * out = (softmax(q k^T * scale) o M / (1 - p)) v with a keep mask M.  M is a pure function of
 * (seed, batch, head, query, key) that the backward regenerates from the same `dropout_p` and `seed`
 * (nothing is stored); dropout_p = 0 is the plain attention above.  The mask comes from a counter-based
 * hash, not from torch's Philox stream: the distribution matches the reference, the individual draws do not. */
int vitk_attention_fwd_dropout(const void* q, const void* k, const void* v, void* out, float* lse,
	int64_t B, int64_t n_head, int64_t Sq, int64_t Skv, int64_t dh,
	int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, float scale, float dropout_p, uint64_t seed, int dtype, void* stream);
int vitk_attention_bwd_dropout(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse,
	void* dq, void* dk, void* dv,
	int64_t B, int64_t n_head, int64_t Sq, int64_t Skv, int64_t dh,
	int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
	float scale, float dropout_p, uint64_t seed, float* scratch, int dtype, void* stream);

/* vitk_attention_bwd_dropout for callers that cannot allocate device memory and have no buffer to hand in as `scratch`
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  The 2 * B * n_head * Sq floats of scratch come from a fixed 8 MiB array inside the library's device image
 * (one copy per GPU, held until the process ends); problems with 2 * B * n_head * Sq > 2,097,152 return
 * VITK_ERR_SCRATCH.  The call allocates and frees nothing.  All calls on one GPU share the array, so calls that may
 * overlap in time (different streams) race; the first call on a device looks the array's address up (do that call
 * outside stream capture).  dropout_p = 0 is the plain backward. */
int vitk_attention_bwd_noalloc(const void* dout, const void* q, const void* k, const void* v, const void* out, const float* lse,
	void* dq, void* dk, void* dv,
	int64_t B, int64_t n_head, int64_t Sq, int64_t Skv, int64_t dh,
	int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
	float scale, float dropout_p, uint64_t seed, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Bandwidth-bound elementwise / reduction kernels.
 */
/* dst[i] = (dst_dtype) src[i] */
int vitk_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);
/* y[i] = x[i] * g[0] with g one float in device memory: the upstream gradient of a scalar loss applied to the stored
 * d(loss)/d(logits) (autograd's backward of nn.CrossEntropyLoss / the DeiT loss, main.py:106-109, deit_trainer.py:30-45)
 * without reading g back to the host. */
int vitk_scale_by(const void* x, const float* g, void* y, int64_t n, int dtype, void* stream);
/* y[c*ldy + r] = x[r*ldx + c] for r < rows, c < cols.  No counterpart op in the reference: it keeps a transposed
 * bf16 copy of a Linear weight so that the input-gradient GEMM of autograd (the dy @ W of nn.Linear's backward,
 * reached from layers/transformers.py:43-54 and layers/attention.py:36-59) reads both operands K-major. */
int vitk_transpose(const void* x, void* y, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy, int dtype, void* stream);
/* All weight copies of a model transposed by ONE launch (after each optimizer step).  `items_dev`: device array of
 * contiguous matrices; `tiles_dev`: device array of n_tiles x 4 int32 {item index, 64-row tile, 64-column tile, 0},
 * one entry per 64x64 tile of every matrix (built once by the caller). */
typedef struct vitk_transpose_item {
	const void* src; /* [rows, cols] contiguous */
	void* dst;       /* [cols, rows] contiguous */
	int64_t rows, cols;
} vitk_transpose_item;
int vitk_transpose_batched(const vitk_transpose_item* items_dev, const int32_t* tiles_dev, int64_t n_tiles, int dtype, void* stream);
/* out[n] (+)= sum_m x[m*ld + n]  — bias gradient of a Linear (autograd of addmm) */
int vitk_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ld, int accumulate, int dtype, void* stream);
/* Gated FFN (layers/transformers.py:48-50): u = [gate | up] with row stride ld_u, a = gate * act(up). */
int vitk_gated_act_fwd(const void* u, void* a, int64_t M, int64_t F, int64_t ld_u, int act, int dtype, void* stream);
/* du = [da * act(up) | da * gate * act'(up)] */
int vitk_gated_act_bwd(const void* da, const void* u, void* du, int64_t M, int64_t F, int64_t ld_u, int act, int dtype, void* stream);
/* y = act(x) and dx = dy * act'(x) as stand-alone kernels (used when no GEMM epilogue is available) */
int vitk_act_fwd(const void* x, void* y, int64_t n, int act, int dtype, void* stream);
int vitk_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int act, int dtype, void* stream);
/* Inverted dropout (nn.Dropout, layers/vit.py:18,23,43,51; layers/transformers.py:42-43,54-55):
 * y[i] = keep(i) ? x[i] / (1 - p) : 0 with keep(i) drawn from a counter-based hash of (seed, i).
 * The backward is the same call on dy with the same seed (the mask is regenerated, never stored).
 * The random stream differs from torch's Philox stream, so training-mode dropout matches the
 * reference in distribution only. */
int vitk_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream);
/* Dropout epoch.  Every mask of this library (vitk_dropout / vitk_dropout_add, the drop_p of vitk_gemm, the dropout_p of
 * the attention calls) is a function of (seed, element); a launch captured into a HIP graph keeps the seed it was captured
 * with, so a replayed training step would draw the same masks every time.  vitk_dropout_epoch_set registers, for the
 * calling thread's current device, one device word (uint32, caller-owned, must outlive every launch made while it is
 * registered) whose value at execution time is mixed into the seed by every later dropout launch on that device; null
 * unregisters.  A step that is captured advances the word once, before its first mask (vitk_dropout_epoch_advance: one
 * tiny kernel on `stream`), so forward and backward of one replay agree and two replays differ.  With nothing
 * registered, or while the word holds 0, the masks are those of the seed alone.  The reference draws torch's Philox
 * stream per call (nn.Dropout, layers/vit.py:17-23), which a captured graph advances the same way. */
int vitk_dropout_epoch_set(const uint32_t* epoch);
int vitk_dropout_epoch_advance(uint32_t* epoch, void* stream);
/* y = res + dropout(dropout(x, p1, seed1), p2, seed2): the branch dropout(s) and the residual add of a block in one
 * pass (layers/vit.py:43-47 and 51-55; the feed-forward branch passes through FeedForward.dropout2,
 * layers/transformers.py:55, and ViTBlock.ff_dropout back to back).  res nullable (then y is the dropped x: the
 * backward of the same expression), p2 = 0 applies one mask only.  Masks as in vitk_dropout. */
int vitk_dropout_add(const void* x, const void* res, void* y, int64_t n, float p1, uint64_t seed1, float p2, uint64_t seed2,
	int dtype, void* stream);
/* y = res + gamma[col] * x (gamma nullable); rows x d.  layers/vit.py:44-47,52-55 when not fused in a GEMM */
int vitk_scale_residual(const void* x, const void* res, const float* gamma, void* y, int64_t rows, int64_t d, int dtype, void* stream);
/* LayerScale backward: dbranch = gamma * dy ; dgamma[col] += sum_rows dy * branch */
int vitk_layerscale_bwd(const void* dy, const void* branch, const float* gamma, void* dbranch, float* dgamma,
	int64_t rows, int64_t d, int dtype, void* stream);
/* Class-token rows (models/vit.py:49-52): x[b, 0, :] = class_token + pos_embed[0]; x is [B, tokens, d]. */
int vitk_cls_rows(void* x, const float* class_token, const float* pos_embed, int64_t B, int64_t tokens, int64_t d, int dtype, void* stream);
/* dpos[t, :] (+)= sum_b dx[b, t, :]  (gradient of the broadcast pos_embed add; row 0 is also d class_token) */
int vitk_batch_sum(const void* dx, float* dpos, int64_t B, int64_t rows, int64_t d, int accumulate, int dtype, void* stream);
/* im2col of non-overlapping patches (layers/embeddings.py:7-16) and its inverse scatter; used by the
 * backward of the patch embedding only — the forward gathers inside the GEMM. */
int vitk_patchify(const void* images, void* patches, int64_t B, int64_t C, int64_t HW, int64_t n_patch, int dtype, void* stream);
int vitk_unpatchify(const void* patches, void* images, int64_t B, int64_t C, int64_t HW, int64_t n_patch, int dtype, void* stream);
/* Copy rows [1, tokens) of every image of x[B, tokens, d] into a dense [B*(tokens-1), d] matrix. */
int vitk_drop_cls_rows(const void* x, void* out, int64_t B, int64_t tokens, int64_t d, int dtype, void* stream);
/* The same for `lead` leading rows per image (class + distillation token): rows [lead, tokens) -> [B*(tokens-lead), d]. */
int vitk_drop_lead_rows(const void* x, void* out, int64_t B, int64_t tokens, int64_t d, int64_t lead, int dtype, void* stream);
/* dst[r * ld_dst + c] += src[r * ld_src + c] for r < rows, c < d: gradients of the class-token rows added into a gradient
 * of all token rows (the out[:, 0] of models/vit.py:63 seen from the last block, whose other output rows nobody reads:
 * autograd's backward of that indexing puts the row gradient into an otherwise zero tensor and adds it). */
int vitk_add_rows(void* dst, const void* src, int64_t rows, int64_t d, int64_t ld_dst, int64_t ld_src, int dtype, void* stream);
/* Token mean pool x[B, S, d] -> y[B, d] and its backward (models/cvt.py:105). */
int vitk_mean_pool_fwd(const void* x, void* y, int64_t B, int64_t S, int64_t d, int dtype, void* stream);
int vitk_mean_pool_bwd(const void* dy, void* dx, int64_t B, int64_t S, int64_t d, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss heads.
 * Cross entropy with label smoothing, mean reduction (main.py:106-109), fused with the DeiT
 * soft-distillation term (deit_trainer.py:30-45):
 *   loss = (1-alpha) * CE(s, y) + alpha * tau^2 * KL(softmax(t/tau) || softmax(s/tau)) / B
 * teacher == null (or alpha == 0) gives plain cross entropy.  Writes loss[0] (float32; the caller
 * zeroes it) and dlogits = d loss / d s in the dtype of the logits.
 */
int vitk_ce_kd_loss(const void* logits, const int64_t* labels, const void* teacher, float* loss, void* dlogits,
	int64_t B, int64_t C, float label_smoothing, float alpha, float tau, int dtype, void* stream);

/* Classifier and distillation heads of a DeiT-style student in one launch (the classifier of models/vit.py:63-65 applied to
 * token row 0, and a second Linear applied to token row 1, the distillation token; deit_trainer.py:23-52 consumes the
 * pair): logits_cls[b, c] = x[b, 0, :] . w_cls[c, :] + b_cls[c],  logits_dist[b, c] = x[b, 1, :] . w_dist[c, :] + b_dist[c].
 * x [B, tokens, d] and the weights [C, d] in `dtype`, biases float32 (nullable), logits [B, C] in `dtype`.  Bandwidth-bound:
 * one wavefront per (head, class), the weight row in registers, lane-strided dot products reduced by shuffles. */
int vitk_dual_head_fwd(const void* x, const void* w_cls, const float* b_cls, const void* w_dist, const float* b_dist,
	void* logits_cls, void* logits_dist, int64_t B, int64_t tokens, int64_t d, int64_t C, int dtype, void* stream);
/* DeiT loss with a distillation head: (1-alpha) * CE(logits_cls, y; label_smoothing) + alpha * tau^2 *
 * KL(softmax(teacher/tau) || softmax(logits_dist/tau)) / B  (deit_trainer.py:30-45 with the soft term moved to the second
 * head).  Writes loss[0] (the caller zeroes it) and both gradients. */
int vitk_deit_dual_loss(const void* logits_cls, const void* logits_dist, const int64_t* labels, const void* teacher, float* loss,
	void* dlogits_cls, void* dlogits_dist, int64_t B, int64_t C, float label_smoothing, float alpha, float tau, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused AdamW over one flat float32 parameter buffer (torch.optim.AdamW semantics, main.py:100):
 * p *= 1 - lr*wd; m,v updated; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).  `wd_mask` (nullable,
 * one byte per element) selects which elements are decayed (utils.py:102-121 param groups).
 * `p_bf16` (nullable) receives a bfloat16 copy of the updated parameters for the next forward.
 * grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).
 */
int vitk_adamw(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, void* p_bf16, int64_t n,
	float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream);
/* The same step with the step count in DEVICE memory (`step_state`: two int64 words, [0] = count, starts at 0, [1] =
 * scratch for the call): the call increments it and derives the bias corrections from it on the device, so a launch captured into a HIP graph advances on every replay
 * (the reference's eager loop, train.py:42-59, re-reads the optimizer's host-side step each iteration). */
int vitk_adamw_device_step(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, void* p_bf16, int64_t n,
	float lr, float beta1, float beta2, float eps, float weight_decay, int64_t* step_state, float grad_scale, void* stream);
/* The same with the hyper-parameters in DEVICE memory too, one set per parameter group of utils.py:102-121:
 * hparams[8] = {lr, weight_decay of the group wd_mask selects, lr, weight_decay of the other group, beta1, beta2, eps, unused}.
 * A learning-rate scheduler (main.py:113-131, schedulers.py:3-35) rewrites them between steps; a launch captured into a
 * HIP graph then follows the schedule on replay. */
int vitk_adamw_device_hparams(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, void* p_bf16, int64_t n,
	const float* hparams, int64_t* step_state, float grad_scale, void* stream);
/* The step of vitk_adamw_device_hparams in pieces, for a caller that pipelines the optimizer under the next forward pass
 * (no counterpart in the reference, whose optimizer.step() in train.py:55 finishes before the next batch starts):
 * vitk_adamw_advance increments the device-side step count once and derives the bias corrections; vitk_adamw_range then
 * updates ANY sub-range of the flat buffers (pointers already offset, n elements) with the corrections and
 * hyper-parameters found in device memory, and with zero_grad != 0 also clears the gradients it has consumed (the
 * optimizer.zero_grad() of train.py:45).  The ranges of one step may run on any stream behind the advance. */
int vitk_adamw_advance(int64_t* step_state, const float* hparams, void* stream);
int vitk_adamw_range(float* p, float* g, float* m, float* v, const uint8_t* wd_mask, void* p_bf16, int64_t n,
	const float* hparams, const int64_t* step_state, float grad_scale, int zero_grad, void* stream);

/* ------------------------------------------------------------------------------------------
 * CvT pieces.
 * Depthwise 3x3 convolution on token-major activations (layers/attention.py:61-62,73-78):
 * x [B, H*W, C] -> y [B, Ho*Wo, C], weight [C, 1, ks, ks] float32 or bf16 as `dtype`, no bias.
 */
int vitk_dwconv_fwd(const void* x, const void* w, void* y, int64_t B, int64_t H, int64_t W, int64_t C,
	int64_t ks, int64_t stride, int64_t pad, int dtype, void* stream);
int vitk_dwconv_bwd(const void* dy, const void* x, const void* w, void* dx, float* dw, int64_t B, int64_t H, int64_t W, int64_t C,
	int64_t ks, int64_t stride, int64_t pad, int dtype, void* stream);
/* The same with row strides (in elements) for the INPUT-sized tensors: x rows are ldx apart, dx rows lddx apart, so that K and V
 * (and their gradients) may be column slices of the joint q | k | v projection buffer of ConvAttention
 * (layers/attention.py:57-64,73-78: three Linear layers on the same input, run here as one GEMM).  The output-sized
 * tensors y and dy stay contiguous [B * Ho * Wo, C]. */
int vitk_dwconv_fwd_ld(const void* x, int64_t ldx, const void* w, void* y, int64_t B, int64_t H, int64_t W, int64_t C,
	int64_t ks, int64_t stride, int64_t pad, int dtype, void* stream);
int vitk_dwconv_bwd_ld(const void* dy, const void* x, int64_t ldx, const void* w, void* dx, int64_t lddx, float* dw, int64_t B,
	int64_t H, int64_t W, int64_t C, int64_t ks, int64_t stride, int64_t pad, int dtype, void* stream);
/* ... and with accumulate_dw != 0 the weight gradient is ADDED into dw (a gradient sink that already holds this step's other
 * contributions; the caller owns its zeroing) instead of overwriting it: saves the fill and the add kernel per ConvAttention. */
int vitk_dwconv_bwd_acc(const void* dy, const void* x, int64_t ldx, const void* w, void* dx, int64_t lddx, float* dw, int accumulate_dw,
	int64_t B, int64_t H, int64_t W, int64_t C, int64_t ks, int64_t stride, int64_t pad, int dtype, void* stream);
/* im2col for the overlapping conv token embedding (layers/embeddings.py:21): x NCHW -> cols [B*Ho*Wo, C*ks*ks]
 * (the convolution itself is vitk_gemm on cols) and col2im (scatter-add) for its input gradient.
 * With tokens_in != 0 the input is token-major [B, H*W, C] (the layout CvT blocks produce), so the
 * 'b (h w) c -> b c h w' rearrange of models/cvt.py:91,98 is folded into the gather.
 * ld_cols >= C*ks*ks is the row stride of cols; im2col writes zeros into the columns past C*ks*ks, so that a K that is
 * not a multiple of 8 (3 * 7 * 7 = 147 in the first CvT stage) reaches the bf16 GEMM already padded. */
int vitk_im2col(const void* x, void* cols, int64_t B, int64_t C, int64_t H, int64_t W, int64_t ks, int64_t stride, int64_t pad,
	int64_t ld_cols, int tokens_in, int dtype, void* stream);
int vitk_col2im(const void* cols, void* dx, int64_t B, int64_t C, int64_t H, int64_t W, int64_t ks, int64_t stride, int64_t pad,
	int64_t ld_cols, int tokens_in, int dtype, void* stream);
/* dst[r, 0:cols] = src[r, 0:cols] converted to dst_dtype, dst[r, cols:ld_dst] = 0: the conv weight [C_out, C*ks*ks]
 * (layers/embeddings.py:21) as the zero-padded K-major operand that goes with a padded cols matrix. */
int vitk_pad_cols(const void* src, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int64_t ld_dst, int src_dtype,
	int dst_dtype, void* stream);
/* BatchNorm over token-major activations x[rows, C] (nn.BatchNorm2d after the conv, layers/embeddings.py:22,26-27;
 * rows = B*Ho*Wo).  training: batch statistics (biased variance for normalisation, unbiased for the running
 * update with `momentum`), save_mean/save_rstd [C] float32 kept for the backward.  eval: running statistics. */
int vitk_batchnorm_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
	void* y, float* save_mean, float* save_rstd, int64_t rows, int64_t C, float eps, float momentum, int training, int dtype, void* stream);
int vitk_batchnorm_bwd(const void* dy, const void* x, const float* gamma, const float* save_mean, const float* save_rstd,
	void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C, int training, int dtype, void* stream);
/* The same with a caller-provided scratch buffer (16-byte aligned, vitk_batchnorm_workspace_bytes(rows, C) bytes or more; the
 * GEMM workspace serves, calls are stream-ordered): the per-row-slab column sums pass through it and are added in a fixed
 * order, so batch statistics and dgamma / dbeta are the same bits from run to run.  The entry points above add them with
 * float atomics in arrival order (so does this one when the buffer is absent or too small). */
int64_t vitk_batchnorm_workspace_bytes(int64_t rows, int64_t C);
int vitk_batchnorm_fwd_ws(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
	void* y, float* save_mean, float* save_rstd, int64_t rows, int64_t C, float eps, float momentum, int training, int dtype,
	void* workspace, int64_t workspace_bytes, void* stream);
int vitk_batchnorm_bwd_ws(const void* dy, const void* x, const float* gamma, const float* save_mean, const float* save_rstd,
	void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C, int training, int dtype, void* workspace,
	int64_t workspace_bytes, void* stream);
/* BatchNorm2d followed by the activation of an MBConv (layers/conv.py:27-37: Conv2d -> BatchNorm2d -> act) in one pass over the
 * tensor: y = act(batchnorm(x)) (act: a VITK_ACT_* code), and its backward from dy = the gradient behind the activation: the
 * normalised value is formed again from x, the saved statistics, gamma and beta, so neither it nor a separate activation pass
 * exists (three fewer passes over the tensor per pair in a training step, one tensor less to keep).  C must be a multiple of 8
 * (bf16) / 4 (f32) and the tensors 16-byte aligned: VITK_ERR_UNSUPPORTED otherwise, nothing launched. */
int vitk_batchnorm_act_fwd(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
	void* y, float* save_mean, float* save_rstd, int64_t rows, int64_t C, float eps, float momentum, int training, int act,
	int dtype, void* workspace, int64_t workspace_bytes, void* stream);
int vitk_batchnorm_act_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* save_mean,
	const float* save_rstd, void* dx, float* dgamma, float* dbeta, int64_t rows, int64_t C, int training, int act, int dtype,
	void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITK_H */
