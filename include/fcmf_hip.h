/* fcmf_hip.h -- C ABI of libfcmf_hip.so: the MI355X (gfx950) kernels underneath the
 * `fcmf_framework` Python module surface.
 *
 * The reference (sonbui25/Multimodal-Aspect-Category-Sentiment-Analysis) has no FFI layer:
 * its operators are whatever torch dispatches from fcmf_framework/{mm_modeling,roi_modeling,
 * fcmf_pretraining,fcmf_multimodal,optimization}.py.  Each entry point below names the
 * reference code (file:line under /root/reference) whose arithmetic it replaces.
 *
 * Conventions
 *  - plain pointers + sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *    synchronises, nothing allocates (workspaces are caller-provided);
 *  - return value: 0 = FCMF_OK, negative = error (no exceptions cross the ABI);
 *  - dtype codes select the ACTIVATION storage type; master parameters, optimizer state,
 *    LayerNorm statistics, logsumexp and gradients of parameters are always float32.
 */
#ifndef FCMF_HIP_H
#define FCMF_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCMF_OK 0
#define FCMF_ERR_ARG (-1)
#define FCMF_ERR_LAUNCH (-2)
#define FCMF_ERR_UNSUPPORTED (-3)
#define FCMF_ERR_COMM (-4)        /* RCCL unavailable or a collective call failed */

#define FCMF_F32 0
#define FCMF_BF16 1
#define FCMF_F64 2

/* GEMM epilogues */
#define FCMF_EPI_NONE 0   /* C = acc + bias                                            */
#define FCMF_EPI_GELU 1   /* C = gelu_erf(acc + bias); aux (if non-NULL) <- acc + bias  */
#define FCMF_EPI_TANH 2   /* C = tanh(acc + bias)                                       */
#define FCMF_EPI_DGELU 3  /* C = acc * gelu'(aux)      (aux = saved pre-activation)     */
#define FCMF_EPI_DTANH 4  /* C = acc * (1 - aux^2)     (aux = saved tanh output)        */
#define FCMF_EPI_ADD 5    /* C = acc + bias + aux      (residual-gradient accumulation)  */

int fcmf_abi_version(void);
/* human-readable "gfx950 ..." build string (host pointer, static storage) */
const char* fcmf_build_info(void);

/* ---------------------------------------------------------------------------------------
 * GEMM context: everything fcmf_gemm uses beyond its arguments.  The library keeps NO mutable process-global state; a
 * context is owned by the caller and used by one host thread / one stream at a time (the Python layer keeps one per
 * (device, stream)).  NULL is a valid context: defaults, no split-K workspace.
 *   set_workspace: caller-owned DEVICE scratch for the k-split partial tiles of weight-gradient GEMMs (accumulate != 0):
 *                  with at least ksplit*M*N*4 bytes the partials are written with plain stores and summed by a reduce
 *                  pass (5x cheaper than 65 536 float atomics per CU); without it, or when it is too small, float
 *                  atomics.  The buffer must stay valid until the stream has drained.  ptr = NULL unregisters.
 *   tune         : benchmark / test knobs; a negative value keeps the current setting.
 *                  force_tile 0 = built-in heuristic, 128 = 128x128 kernel, 256 / 192 = persistent 256x256 / 192x256 kernel
 *                  wherever its preconditions hold; kb 32 / 64 = depth of the k-tiles of the K-contiguous persistent kernels
 *                  (both depths issue the same MFMAs in the same order: bit-identical results); num_cus = workgroups of
 *                  the persistent kernels (8..256; leaves CUs to kernels that run beside them, e.g. RCCL);
 *                  nt_min_bytes = bf16 outputs of at least this size leave with nontemporal stores (default 0 = all).
 *   last_kernel  : name of the kernel the context's last fcmf_gemm dispatched, e.g.
 *                  "gemm_bf16_tile256_kernel<0,1,bf16,GELU>" (benchmarks attribute launch time by it; host string owned
 *                  by the context). */
typedef struct fcmf_gemm_ctx fcmf_gemm_ctx;
int fcmf_gemm_ctx_create(fcmf_gemm_ctx** ctx);
int fcmf_gemm_ctx_destroy(fcmf_gemm_ctx* ctx);
int fcmf_gemm_ctx_set_workspace(fcmf_gemm_ctx* ctx, void* ptr, int64_t bytes);
int fcmf_gemm_ctx_tune(fcmf_gemm_ctx* ctx, int force_tile, int kb, int num_cus, int64_t nt_min_bytes);
const char* fcmf_gemm_ctx_last_kernel(const fcmf_gemm_ctx* ctx);

/* ---------------------------------------------------------------------------------------
 * GEMM:  C[M,N] (+)= epilogue( op(A)[M,K] * op(B)[K,N] + bias[N] )
 *   ctx: GEMM context (above) or NULL.
 *   trans_a = 0: A stored [M,K] (row stride lda)      trans_a = 1: A stored [K,M]
 *   trans_b = 0: B stored [N,K] (nn.Linear weight)    trans_b = 1: B stored [K,N]
 *   in_dtype: dtype of A and B; out_dtype: dtype of C and aux; bias is float32 or NULL.
 *   accumulate != 0 (out_dtype must be F32): C += result (k-split partials go through the context's workspace, or are
 *   added with float atomics when there is none -- C must be initialised either way).
 *   colsum (float32 [N], may be NULL; not with accumulate): colsum[n] += sum_m C[m,n] -- the bias
 *   gradient of the layer that produced the operand, fused into the epilogue.
 * Replaces nn.Linear forward/backward everywhere on the path: mm_modeling.py:182-184,
 * 229-231,272,308,320,422 ; fcmf_pretraining.py:25-26 ; roi_modeling.py:73 ;
 * fcmf_multimodal.py:18 and their autograd (dX = dY*W, dW = dY^T*X).
 * bf16 inputs whose contiguous dimensions are multiples of 8 elements with 16-byte aligned
 * bases run on the MFMA bf16 kernel; everything else runs on the generic f32-MFMA kernel. */
int fcmf_gemm(fcmf_gemm_ctx* ctx, const void* A, const void* B, void* C, const float* bias, void* aux, float* colsum,
              int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
              int trans_a, int trans_b, int in_dtype, int out_dtype,
              int epilogue, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------
 * FP8 (OCP e4m3fn) path of BASELINE configs[4] (FCMF-large): forward and dX GEMMs on the block-scaled matrix instruction
 * v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16 MFMA rate), float32 accumulation; weight gradients stay bf16.
 * The only "large" hook of the reference is the constant block of mm_modeling.py:21-30; the arithmetic replaced is the
 * same nn.Linear forward / dX as fcmf_gemm.
 *   fcmf_quant_fp8_rows: x [rows, K] (FCMF_BF16 / FCMF_F32, row stride ldx elements) -> q [rows, K] e4m3 bytes (row stride
 *       ldq bytes) with one dequantisation scale per row: scale[r] = max|x[r, :]| / 448 (1 for a zero row),
 *       q = rne(x / scale[r]).  K, ldx, ldq multiples of 8.
 *   fcmf_gemm_fp8: C[M,N] bf16 = epilogue( (A_q[M,K] * B_q[N,K]^T) * sa[m] * sb[n] + bias[n] ); both operands K-contiguous
 *       e4m3 with their row scales (A: activations / output gradients quantised per token; B: the weight quantised per output
 *       row -- or its transposed copy per input row for dX).  Epilogues NONE / GELU / DGELU / ADD, aux and colsum as in
 *       fcmf_gemm.  K % 128 == 0, lda / ldb % 16 == 0 (bytes), N and ldc % 8 == 0, M, N >= 256: otherwise
 *       FCMF_ERR_UNSUPPORTED (the caller keeps the bf16 path). */
int fcmf_quant_fp8_rows(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows, int K, int dtype, void* stream);
int fcmf_gemm_fp8(fcmf_gemm_ctx* ctx, const void* A, const float* sa, const void* B, const float* sb, void* C,
                  const float* bias, void* aux, float* colsum, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                  int epilogue, void* stream);

/* column sums: out[n] (+)= sum_m X[m,n]  (bias gradients).  X dtype = dtype, out float32. */
int fcmf_colsum(const void* X, float* out, int M, int N, int64_t ldx, int dtype,
                int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------
 * Two-segment multi-head attention (VALU, any dtype).  One "group" g has R query rows; every
 * row attends to T1 keys shared by the group (K1/V1) followed by T2 keys private to the row
 * (K2/V2, indexed by g / group_div so that several groups can share the private segment).
 *   score[t] = scale * <q, k_t> + mask[g, t] + bias[g / group_div, h, r, t]
 *   causal != 0: score[t] = -1e4 where t > r   (IAOG `Attention`, mm_modeling.py:115-124)
 *   p = dropout(softmax(score)); out = sum_t p[t] v_t ; lse[g,h,r] = logsumexp(score)
 * Layouts (elements): Q  (g,r,h,:) at q  + g*q_sg  + r*q_sr  + h*d
 *                     K1 (g,t,h,:) at k1 + g*k1_sg + t*k1_st + h*d          (V1 same strides)
 *                     K2 (g2,r,t,h,:) at k2 + g2*k2_sg + r*k2_sr + t*k2_st + h*d  (V2 same)
 *                     O  (g,r,h,:) at o  + g*o_sg  + r*o_sr  + h*d          (dO same)
 * head_quirk != 0 reproduces mm_modeling.py:79-85: output slot h of group g READS Q/K/V head
 * (h*G + g) % heads; the backward WRITES dq/dk/dv at slot h (several slots may read one head, so
 * the caller scatter-adds slots back to heads).
 * Replaces BertSelfAttention/BertCoAttention (mm_modeling.py:193-219,240-266), HF
 * eager_attention_forward, box_attention (roi_modeling.py:14-47) and Attention
 * (mm_modeling.py:66-132). */
typedef struct {
  int dtype, G, heads, d, R, T1, T2, group_div;
  int64_t q_sg, q_sr, k1_sg, k1_st, k2_sg, k2_sr, k2_st, o_sg, o_sr;
  const void *q, *k1, *v1, *k2, *v2;
  const float* mask;   /* [G, T1+T2] additive or NULL */
  const float* bias;   /* [G/group_div, heads, R, T1+T2] additive or NULL */
  float scale, dropout_p;
  uint64_t seed;
  int causal, head_quirk;
} fcmf_attn_desc;

int fcmf_attn_small_fwd(const fcmf_attn_desc* desc /*host*/, void* out, float* lse, void* stream);
/* Gradients are written DENSE (independent of the input strides) and fully overwritten:
 *   dq  [ceil(T1/128) or 1, G, R, heads*d] -- one partial per 128-key chunk of the shared segment;
 *       the caller sums the leading axis (fcmf_sum_axis);
 *   dk1/dv1 [G, T1, heads*d];   dk2/dv2 [G, R, T2, heads*d] (indexed by g, NOT g/group_div: the
 *       caller sums groups that share a private segment);
 *   dbias (float32, may be NULL) [G, heads, R, T1+T2].
 * When v1 == k1 (IAOG) pass dv1 = NULL: dk1 receives both terms.  Limits: T1+T2 <= 512, T2 <= 128,
 * d <= 128 (FCMF-large: 256 text keys + 100 ROI keys in the shared mm_attention layer). */
int fcmf_attn_small_bwd(const fcmf_attn_desc* desc /*host*/, const void* out, const void* dout,
                        const float* lse, void* dq, void* dk1, void* dv1, void* dk2, void* dv2,
                        float* dbias, void* stream);
/* The same backward where the `group_div` consecutive groups that share private keys (the six aspects of a review:
 * fcmf_multimodal.py:84-124 under aspect batching) also share their gradients: dk2 / dv2 are
 * [G/group_div, R, T2, heads*d], already summed over the group.  scratch: caller-owned device memory,
 * >= 2*G*heads*R*T2*4 bytes (the private keys' probabilities and score gradients between the two kernels).
 * group_div <= 8, G % group_div == 0, no head_quirk; FCMF_ERR_UNSUPPORTED otherwise (use fcmf_attn_small_bwd and sum). */
int fcmf_attn_small_bwd_grouped(const fcmf_attn_desc* desc, const void* out, const void* dout,
                                const float* lse, void* dq, void* dk1, void* dv1, void* dk2, void* dv2,
                                float* dbias, float* scratch, int64_t scratch_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * MFMA self/cross attention for bf16, head dim 64, Tq, Tk <= 256 (the text-encoder layers: 128 tokens in FCMF-base,
 * 256 in FCMF-large; the backward handles every query of a (sequence, head) in one workgroup, so Tq <= 256 there too):
 *   Q [G,Tq,heads*64], K/V [G,Tk,heads*64] (row strides ldq/ldk elements), mask [G,Tk] additive.
 * Replaces HF RobertaSelfAttention + eager_attention_forward and mm_modeling.py:193-219. */
int fcmf_attn_mfma_fwd(const void* q, const void* k, const void* v, const float* mask,
                       void* out, float* lse, int G, int heads, int Tq, int Tk,
                       int64_t ldq, int64_t ldk, int64_t ldo, float scale,
                       float dropout_p, uint64_t seed, void* stream);
int fcmf_attn_mfma_bwd(const void* q, const void* k, const void* v, const float* mask,
                       const void* out, const void* dout, const float* lse,
                       void* dq, void* dk, void* dv, int G, int heads, int Tq, int Tk,
                       int64_t ldq, int64_t ldk, int64_t ldo, float scale,
                       float dropout_p, uint64_t seed, float* colsum, void* stream);
/* colsum (may be NULL): float32 [G][3*heads*64]; row g receives the column sums of sequence g's dq | dk | dv (taken from
 * the f32 accumulators).  Their sum over g is the bias gradient of a fused q|k|v projection (the autograd of
 * mm_modeling.py:182-184 / HF RobertaSelfAttention's three nn.Linear biases) without another pass over dqkv. */

/* ---------------------------------------------------------------------------------------
 * y = LayerNorm(dropout(x) + res) * gamma + beta   (eps inside the sqrt, biased variance)
 * Replaces BertSelfOutput/BertOutput/AddNorm + FCMFLayerNorm (mm_modeling.py:158-171,
 * 276-280,324-328,566-573) and HF nn.LayerNorm(eps=1e-5).
 *   x [rows,H] dense; res row r at res + r*res_stride (NULL = no residual);
 *   z (optional, may alias x) <- dropout(x)+res ; mean/rstd float32 [rows]. */
int fcmf_add_ln_fwd(const void* x, const void* res, int64_t res_stride, const float* gamma,
                    const float* beta, void* y, void* z, float* mean, float* rstd,
                    int rows, int H, float eps, float dropout_p, uint64_t seed, int dtype,
                    void* stream);
/* dz [rows,H] <- grad wrt (dropout(x)+res); dx (NULL when dropout_p==0: dx == dz) <- grad wrt x;
 * dgamma/dbeta float32 [H] are ACCUMULATED (atomics); dxsum (float32 [H], may be NULL) accumulates
 * the column sums of dx = the bias gradient of the Linear that produced x. */
int fcmf_add_ln_bwd(const void* dy, const void* z, const float* gamma, const float* mean,
                    const float* rstd, void* dz, void* dx, float* dgamma, float* dbeta, float* dxsum,
                    float* workspace, int rows, int H, float dropout_p, uint64_t seed, int dtype,
                    void* stream);
/* fcmf_add_ln_fwd / fcmf_add_ln_bwd that ALSO emit the e4m3 copy (q8 [rows, H] bytes + qscale [rows], as fcmf_quant_fp8_rows
 * would produce them from the stored values) of the tensor the next fp8 GEMM consumes: y for the forward, the gradient that
 * flows into the producing Linear (dx with dropout, else dz) for the backward -- no separate quantisation pass over it. */
int fcmf_add_ln_fwd_fp8(const void* x, const void* res, int64_t res_stride, const float* gamma, const float* beta, void* y,
                        void* z, float* mean, float* rstd, int rows, int H, float eps, float dropout_p, uint64_t seed,
                        int dtype, void* q8, float* qscale, void* stream);
int fcmf_add_ln_bwd_fp8(const void* dy, const void* z, const float* gamma, const float* mean, const float* rstd, void* dz,
                        void* dx, float* dgamma, float* dbeta, float* dxsum, float* workspace, int rows, int H,
                        float dropout_p, uint64_t seed, int dtype, void* q8, float* qscale, void* stream);
/* floats of scratch `workspace` must hold (per-workgroup partial column sums, reduced by a second
 * kernel without atomics); workspace == NULL falls back to float atomics. */
int64_t fcmf_add_ln_bwd_workspace(int rows, int H);

/* ---------------------------------------------------------------------------------------
 * RoBERTa embeddings (HF RobertaEmbeddings via mm_modeling.py:440-446).
 * position ids = cumsum(ids != pad) * (ids != pad) + pad, one sequence per wave. */
int fcmf_position_ids(const int64_t* ids, int64_t* pos, int nseq, int S, int pad_id, void* stream);
/* z = word[ids] + pos_table[pos] + type[type_ids] (tables float32), then LayerNorm as above. */
int fcmf_embed_ln_fwd(const int64_t* ids, const int64_t* pos, const int64_t* type_ids,
                      const float* word, const float* pos_table, const float* type_table,
                      const float* gamma, const float* beta, void* y, void* z, float* mean,
                      float* rstd, int ntok, int H, float eps, float dropout_p, uint64_t seed,
                      int dtype, void* stream);
/* scatter-add dz into the float32 table gradients (pad rows of word/pos receive nothing).  dpos may be NULL when
 * the position table is handled by fcmf_embed_pos_bwd. */
int fcmf_embed_bwd(const void* dz, const int64_t* ids, const int64_t* pos, const int64_t* type_ids,
                   float* dword, float* dpos, float* dtype_table, int ntok, int H, int pad_id,
                   int dtype, void* stream);
/* position-table gradient for tokens laid out [nseq, S]: same result as the dpos part of fcmf_embed_bwd (ACCUMULATED
 * into dpos), summed per sequence offset in registers first -- the offsets of all sequences share position rows. */
int fcmf_embed_pos_bwd(const void* dz, const int64_t* pos, float* dpos, int nseq, int S, int H, int pad_id,
                       int dtype, void* stream);
/* the position-table AND the type-table gradient (both ACCUMULATED) of tokens laid out [nseq, S] in one pass over dz; call
 * fcmf_embed_bwd with dpos = dtype_table = NULL beside it.  FCMF_ERR_UNSUPPORTED (H not a multiple of 16 bytes of columns or
 * more than 256 such groups, unaligned dz): use fcmf_embed_bwd's type-table path + fcmf_embed_pos_bwd. */
int fcmf_embed_pos_type_bwd(const void* dz, const int64_t* pos, const int64_t* type_ids, float* dpos,
                            float* dtype_table, int nseq, int S, int H, int pad_id, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Box geometry (roi_modeling.py:79-138,161-163 and the log-clamp of :40), fused:
 *   bias[g,h,i,j] = log(max(relu(<WG_h, emb(box_i, box_j)> + b_h), 1e-6))
 * coords [G,N,4] = (x_min,x_max,y_min,y_max) in float64 or float32 (coord_dtype);
 * wg_w [heads,64], wg_b [heads] float32 (heads <= 8); bias float32 [G,heads,N,N];
 * dim_mat: device float32[8] = 1/1000^(k/8) as the reference rounds it (roi_modeling.py:123-125).
 * coord_dtype = FCMF_F32 | FCMF_BOX_FAST_TRIG: the 64 sines / cosines of a box pair on the hardware's v_sin_f32 / v_cos_f32
 * (arguments up to ~110 revolutions: absolute error <= 1e-4, against 1e-7 of sincosf) -- for a bias that is consumed in bf16. */
#define FCMF_BOX_FAST_TRIG 0x100
int fcmf_box_bias_fwd(const void* coords, int coord_dtype, const float* dim_mat, const float* wg_w,
                      const float* wg_b, float* bias, int G, int N, int heads, void* stream);
/* dwg_w/dwg_b are ACCUMULATED. */
int fcmf_box_bias_bwd(const void* coords, int coord_dtype, const float* dim_mat, const float* wg_w,
                      const float* wg_b, const float* dbias, float* dwg_w, float* dwg_b, int G, int N,
                      int heads, void* stream);
/* the raw 64-d relational embedding [G,N,N,64] float32 (tests / API parity of
 * BoxMultiHeadedAttention.BoxRelationalEmbedding). */
int fcmf_box_embedding(const void* coords, int coord_dtype, const float* dim_mat, float* emb, int G,
                       int N, void* stream);

/* ---------------------------------------------------------------------------------------
 * Cross entropy over C classes (run_multimodal_fcmf.py:290,474; ignore_index for IAOG
 * run_pretraining_fcmf.py:322-324).  loss_rows[n] = -log softmax(logits[n])[label] (0 if ignored);
 * nvalid <- number of non-ignored rows (float). */
int fcmf_xent_fwd(const void* logits, int64_t ld, const int64_t* labels, float* loss_rows,
                  float* nvalid, int n, int C, int64_t ignore_index, int dtype, void* stream);
/* dlogits[n,c] = (softmax - onehot) * (*scale_ptr) * extra_scale  (0 for ignored rows) */
int fcmf_xent_bwd(const void* logits, int64_t ld, const int64_t* labels, void* dlogits,
                  int64_t ldd, const float* scale_ptr, float extra_scale, int n, int C,
                  int64_t ignore_index, int dtype, void* stream);

/* The reduction of CrossEntropyLoss(reduction="mean") (run_multimodal_fcmf.py:290,474) over fcmf_xent_fwd's rows, times
 * `mult` (the driver sums the 6 aspects' batch means, :463-475 = mean over all rows x 6), in one fixed-order pass:
 * out2[0] = mult * sum(loss_rows) / #(labels != ignore_index), out2[1] = mult / #(...) -- the scale of fcmf_xent_bwd. */
int fcmf_xent_mean(const float* loss_rows, const int64_t* labels, int n, int64_t ignore_index,
                   float mult, float* out2, void* stream);

/* ---------------------------------------------------------------------------------------
 * Additive attention mask of a 0 / 1 int64 mask [rows, >= cols] with row stride ld (elements):
 * out[r][c] = (1 - mask[r][c]) * value as float32 [rows, cols]  (fcmf_pretraining.py:53-56,97-100,133-136: value = -10000;
 * HF get_extended_attention_mask for the text encoder: value = finfo(float32).min). */
int fcmf_additive_mask(const int64_t* mask, int64_t ld, float* out, int rows, int cols, float value,
                       void* stream);

/* ---------------------------------------------------------------------------------------
 * elementwise helpers */
int fcmf_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);
/* dst (bfloat16 [cols, rows]) = transpose(src float32 [rows, cols]): transposed weight copies for the dX GEMMs */
int fcmf_cast_transpose(const float* src, void* dst, int rows, int cols, void* stream);
/* the same for many weights in ONE launch (all transposed copies go stale together at the optimizer step):
 * src_ptrs / dst_ptrs: device int64 arrays of device addresses (float32 [rows_t, cols_t] -> bfloat16 [cols_t, rows_t]);
 * dims: device int32 [ntensors][2] = (rows, cols); tile_desc: device int32 [ntiles][3] = (tensor, tile row, tile column),
 * one 64x64 tile per workgroup. */
int fcmf_multi_cast_transpose(const int64_t* src_ptrs, const int64_t* dst_ptrs, const int32_t* dims,
                              const int32_t* tile_desc, int ntiles, void* stream);
/* y = x * dropout_mask(seed) / (1-p) ; used for the classifier-head dropout
 * (fcmf_multimodal.py:49) and its backward (same call on dy). */
int fcmf_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream);
/* out = dy * act'(aux): kind 0 = tanh (aux = tanh OUTPUT, BertPooler mm_modeling.py:430),
 * kind 1 = erf-GELU (aux = PRE-activation, mm_modeling.py:15). */
int fcmf_act_bwd(const void* dy, const void* aux, void* out, int64_t n, int kind, int dtype,
                 void* stream);
/* out[i] (+)= sum over the `reps` copies: in is [outer, reps, inner] -> out [outer, inner] */
int fcmf_sum_axis(const void* in, void* out, int64_t outer, int reps, int64_t inner, int dtype,
                  void* stream);

/* IAOG decoder glue (FCMFSeq2Seq / IAOGDecoder):
 *  fcmf_embed_scale_fwd: out[row] = weight[ids[row]] * scale + pos_table[row % S]  -- `self.embedding(X) * sqrt(H)` followed by
 *      PositionalEncoding's `X + P[:, :T]` (mm_modeling.py:650 and :619-633; pos_table may be NULL); weight / pos_table float32,
 *      out in `dtype`; V = rows of `weight`: an id outside [0, V) reads nothing and makes its output row NaN (torch's gather
 *      raises a device assert there; a NaN loss is the stream-ordered way of being as loud);
 *  fcmf_embed_scale_bwd: dweight[ids[row]] += dy[row] * scale (the embedding gradient; dweight float32 [V, H], zero-initialised by the
 *      caller); rows with an id outside [0, V) are skipped -- dweight is a slice of the flat gradient arena, its neighbour is another
 *      parameter's gradient -- and counted in *oob_count (device int32, may be NULL);
 *  fcmf_head_gather: the decoder `Attention` pairs output slot s of batch element g with head (s*G + g) % heads
 *      (mm_modeling.py:79-85).  slot [G, T, heads*d] dense (gradients per slot, as fcmf_attn_small_bwd with head_quirk writes
 *      them) -> out[g, t, h*d + j] = sum of the slots that read head h; out rows have stride ldo elements (so that dk | dq of a
 *      fused projection land side by side). */
int fcmf_embed_scale_fwd(const int64_t* ids, const float* weight, const float* pos_table, void* out, int n, int H, int S,
                         int64_t V, float scale, int dtype, void* stream);
int fcmf_embed_scale_bwd(const void* dy, const int64_t* ids, float* dweight, int n, int H, int64_t V, int32_t* oob_count,
                         float scale, int dtype, void* stream);
int fcmf_head_gather(const void* slot, void* out, int64_t ldo, int G, int T, int heads, int d, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimizer (run_multimodal_fcmf.py:485-488: clip_grad_norm_ + torch.optim.AdamW, 4 groups).
 * Tensors are described by device tables: ptr tables are int64 device arrays of device
 * addresses; `chunk_tensor[c]`/`chunk_offset[c]` assign chunk c (chunk_size elements) to a
 * tensor and a start offset. */
int fcmf_multi_sumsq(const int64_t* g_ptrs, const int64_t* sizes, const int32_t* chunk_tensor,
                     const int64_t* chunk_offset, int nchunks, int chunk_size,
                     double* sumsq /* [1], accumulated */, double* per_tensor /* [ntensors] or NULL */,
                     void* stream);
/* AdamW with decoupled weight decay, bias correction, eps outside sqrt(v_hat) (torch
 * semantics).  The clip coefficient min(1, max_norm/(sqrt(*sumsq)+1e-6)) is computed on
 * device (max_norm <= 0 disables).  group_of[t] selects lr[]/wd[] (host arrays, <= 8 groups).
 * bf16_ptrs (may be NULL; entries may be 0): shadow bf16 copies refreshed after the update. */
int fcmf_multi_adamw(const int64_t* p_ptrs, const int64_t* g_ptrs, const int64_t* m_ptrs,
                     const int64_t* v_ptrs, const int64_t* bf16_ptrs, const int64_t* sizes,
                     const int32_t* group_of, const int32_t* chunk_tensor,
                     const int64_t* chunk_offset, int nchunks, int chunk_size,
                     const float* lr /*host[8]*/, const float* wd /*host[8]*/, int ngroups,
                     float beta1, float beta2, float eps, int step, const double* sumsq,
                     float max_norm, void* stream);
/* BertAdam.step for one tensor (optimization.py:94-162): per-parameter clip, no bias
 * correction, weight decay added to the update; `lr_scheduled` is computed by the host
 * from the warmup schedule. `scratch` is a device double[1]. */
int fcmf_bertadam(float* p, const float* g, float* m, float* v, int64_t n, float lr_scheduled,
                  float beta1, float beta2, float eps, float weight_decay, float max_grad_norm,
                  double* scratch, void* stream);

/* ---------------------------------------------------------------------------------------
 * ResNet-152 trunk (fcmf_framework/resnet_utils.py:13-30,39-56 driving torchvision's resnet152:
 * conv1/bn1/relu/maxpool/layer1..4, then adaptive_avg_pool2d / global mean).  Activations are NHWC
 * ([N*H*W, C] row-major) so every convolution is an fcmf_gemm: 1x1 stride-1 convolutions directly on the
 * activation matrix, all others on the patch matrix written by fcmf_conv_im2col; weights are passed as
 * [Cout, kh*kw*Cin] with k = (r, s, c).
 *
 * fcmf_conv_im2col: dst[row, (r*kw + s)*C + c] = src(n, ho*stride + r - pad, wo*stride + s - pad, c) or 0,
 * row = (n*Ho + ho)*Wo + wo, Ho = (H + 2*pad - kh)/stride + 1; columns kh*kw*C .. Kpad-1 are zero (the stem's
 * K = 147 is padded to a multiple of 32).  src element (n,h,w,c) at n*sn + h*sh + w*sw + c*sc: NCHW float32
 * crops and NHWC activations alike.  (nn.Conv2d's input gather, torchvision Bottleneck conv2 / downsample.0 / conv1) */
int fcmf_conv_im2col(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int H, int W, int C,
                     int64_t sn, int64_t sh, int64_t sw, int64_t sc, int kh, int kw, int stride, int pad,
                     int Kpad, void* stream);
/* GROUPED BatchNorm2d batch statistics.  The reference calls the trunk once per image index / per (image, ROI)
 * with B crops each (run_multimodal_fcmf.py:449-457) in train() mode (:431): rows are packed group-major, group g
 * = rows [g*rows_per_group, (g+1)*rows_per_group), and every group gets ITS OWN statistics.
 * sums: double workspace of fcmf_bn_stats_workspace(rows_per_group, groups, C) elements, fully overwritten with
 * per-(group, row chunk, channel) partial (sum, sum of squares) pairs -- no atomics -- followed by their per-(group,
 * channel) totals (a second, tree-reduction kernel); fcmf_bn_finalize reads the totals. */
int64_t fcmf_bn_stats_workspace(int64_t rows_per_group, int groups, int C);
int fcmf_bn_stats(const void* x, double* sums, int64_t rows_per_group, int groups, int C, int dtype, void* stream);
/* sums != NULL (training; as written by fcmf_bn_stats with rows_per_group == count): scale/shift [groups, C] from
 * each group's batch mean and biased variance (y = x*scale + shift == (x-mean)/sqrt(var+eps)*gamma + beta);
 * running_mean / running_var receive one momentum update per group in group order with the unbiased variance
 * (nn.BatchNorm2d over `groups` calls).  sums == NULL (eval): scale/shift [C] from the running statistics.
 * mean_out / rstd_out (both or neither; [groups, C], eval: [C]): the statistics the backward needs. */
int fcmf_bn_finalize(const double* sums, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out, int C,
                     int groups, int64_t count, float momentum, float eps, void* stream);
/* y = relu?(x * scale[g] + shift[g] (+ res)), g = row / rows_per_group (eval: rows_per_group = rows); y may alias x.
 * (bn -> relu, and the bottleneck's bn3 -> += identity -> relu) */
int fcmf_bn_apply(const void* x, const void* res, void* y, const float* scale, const float* shift, int64_t rows,
                  int C, int64_t rows_per_group, int relu, int dtype, void* stream);
/* the same with y an NHWC tensor that carries a zero border of `pad` pixels ([n, H + 2 pad, W + 2 pad, C], border zeroed once by
 * its owner; rows = n * H * W): the input of an implicit-GEMM 3x3 convolution, produced without a padding pass */
int fcmf_bn_apply_pad(const void* x, const void* res, void* y_padded, const float* scale, const float* shift,
                      int64_t rows, int C, int64_t rows_per_group, int relu, int H, int W, int pad, int dtype, void* stream);
/* fcmf_bn_finalize + fcmf_bn_apply / fcmf_bn_apply_pad in ONE launch: every workgroup derives the scale / shift of its channels
 * from the totals in `sums` itself (sums == NULL: eval, running statistics), the first workgroup of a group leaves mean / rstd
 * [groups, C] (both or neither) for the backward, workgroup 0 applies the `groups` running-statistics updates in group order.
 * rows = groups * rows_per_group; pad = 0: y [rows, C] (may alias x); pad > 0: y the zero-bordered NHWC buffer (rows = n * H * W).
 * FCMF_ERR_UNSUPPORTED unless C / (16 / sizeof(T)) is a power of two <= 256 and x / y / res are 16-byte aligned (the caller then
 * takes the two separate entry points). */
int fcmf_bn_finalize_apply(const void* x, const void* res, void* y, const double* sums, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float* mean_out, float* rstd_out, int C, int groups,
                           int64_t rows_per_group, float momentum, float eps, int relu, int H, int W, int pad, int dtype, void* stream);
/* Implicit-GEMM convolution (no patch matrix): y[(n, oy, ox), co] = sum_{ky,kx,c} x[n, oy*stride + ky, ox*stride + kx, c] * w[co, (ky,kx,c)]
 * on the bf16 MFMA GEMM kernels -- the LDS-DMA of a k-tile reads tap (ky, kx) of every output pixel's receptive field straight
 * from the NHWC activation x [n, Hp, Wp, C], which INCLUDES the zero border where the convolution pads (Hp = H + 2 pad).
 * ResNet-152 trunk: every 3x3 convolution and the strided 1x1 shortcuts (resnet_utils.py:13-24 driving torchvision's Bottleneck);
 * bf16, C a power of two >= 64, w [Cout, kh*kw*C] row-major in (ky, kx, c) order, y [n*Ho*Wo, Cout]. */
int fcmf_conv_gemm(fcmf_gemm_ctx* ctx, const void* x, const void* w, void* y, int n, int Hp, int Wp, int C, int Ho, int Wo,
                   int kh, int kw, int stride, int Cout, void* stream);
/* f32 C (+)= op(A) op(B) of bf16 operands with a COLUMN-BLOCKED output: element (i, j) is stored at
 *   C + (j / col_block) * col_block_stride + i * ldc + j % col_block          (col_block % 4 == 0, N % col_block == 0).
 * The per-head projection weights of the IAOG decoder's Attention are [n_head, E, d] tensors (mm_modeling.py:57-58); their gradient
 * dW^T [E, n_head * d] = x^T dY written with col_block = d, col_block_stride = E * d, ldc = d lands in exactly that layout (the
 * autograd of the reference's repeat + bmm, mm_modeling.py:79-92, without a permute copy per parameter).  Split-K through the
 * context's workspace only (the reduce pass writes the blocked layout): FCMF_ERR_UNSUPPORTED otherwise -- callers fall back to
 * fcmf_gemm into a plain buffer. */
int fcmf_gemm_colblocks(fcmf_gemm_ctx* ctx, const void* A, const void* B, float* C, int M, int N, int K, int64_t lda,
                        int64_t ldb, int64_t ldc, int trans_a, int trans_b, int col_block, int64_t col_block_stride,
                        int accumulate, void* stream);

/* `count` same-shape weight gradients in one launch: C_i [M, N] float32 (+)= A_i^T B_i with A_i = dY_i [K, M], B_i = X_i [K, N]
 * (bf16, K = tokens) -- nn.Linear's weight gradient (loss.backward(), run_multimodal_fcmf.py:466) for the layers of an encoder,
 * which the host queues during the backward pass and multiplies together: the tiles of all matrices form ONE work list for the
 * persistent 256 x 256 kernel (a 768 x 768 gradient alone is 9 tiles; alone it fills the chip only through a 28-way split of K
 * and a 64 MB partial-tile round trip).  A / B / C: host arrays of `count` device pointers.  accumulate as fcmf_gemm.  Shapes the
 * persistent kernel does not take (M or N < 256, unaligned) and count == 1 run as `count` fcmf_gemm calls: same results. */
int fcmf_gemm_dw_batched(fcmf_gemm_ctx* ctx, int count, const void* const* A, const void* const* B, void* const* C, int M, int N, int K,
                         int64_t lda, int64_t ldb, int64_t ldc, int accumulate, void* stream);
/* Implicit-GEMM convolution for inputs with few channels -- the trunk's stem, conv1 = 7x7 / stride 2 / pad 3 on RGB crops
 * (torchvision resnet152.conv1 driven by resnet_utils.py:13-24), whose patch matrix was 1.8 GB per 448 crops: x is NHWC bf16 with
 * `pix` elements per pixel (a power of two: 4 = RGB0, written by fcmf_pack_rgb0) and its zero border; for every kernel row ky the
 * contraction walks ONE contiguous run of `run` elements (a power of two >= 32, >= kw * pix: 32 = 8 pixels for kw = 7) starting at
 * pixel (oy*stride + ky, ox*stride).  K = kh * run; w [Cout, kh * run] bf16 holds zeros where the run exceeds the kernel (kx >= kw,
 * the padding channel).  stride * pix must be a multiple of 8 (16-byte DMA).  y [n*Ho*Wo, Cout] bf16.  stats (optional, may be
 * NULL): the block statistics of y as fcmf_gemm_colstats writes them (same restriction: FCMF_ERR_UNSUPPORTED for Cout < 256). */
int fcmf_conv_gemm_runs(fcmf_gemm_ctx* ctx, const void* x, const void* w, void* y, float* stats, int n, int Hp, int Wp, int pix, int run,
                        int Ho, int Wo, int kh, int stride, int Cout, void* stream);
/* crops in any layout (element strides of n, h, w, c; float32 / float64 / bf16; 3 channels) -> the interior of dst
 * [N, H + 2 pad, Wp, 4] bf16, Wp >= W + 2 pad (channel 3 = 0; the border is NOT written: zero it once) */
int fcmf_pack_rgb0(const void* src, int src_dtype, void* dst, int N, int H, int W, int64_t sn, int64_t sh, int64_t sw, int64_t sc,
                   int pad, int Wp, void* stream);
/* The 1x1 (plain GEMM: A [M, K] = the NHWC activation, W [N, K]) and implicit-GEMM convolutions with the statistics of the
 * train-mode BatchNorm that follows EVERY convolution of the trunk (torchvision Bottleneck: conv -> bn, resnet_utils.py:13-24) as a
 * by-product of the epilogue: stats [ceil(M / R)][N][2] float32 <- (sum, sum of squares) of each block of R output rows per
 * output channel, of the values as stored (bf16-rounded); plain stores, deterministic.  bf16 only.  R =
 * fcmf_gemm_colstats_block_rows(ctx, M, N, K): 128 for the 256-column tiles, 256 for the narrow layouts (32 <= N <= 128 with
 * M >= 8192 and K % 64 == 0), 0 where no kernel emits statistics.  FCMF_ERR_UNSUPPORTED where
 * the shape does not run on the 256-row persistent kernel (M or N < 256, unaligned): run fcmf_gemm / fcmf_conv_gemm +
 * fcmf_bn_stats instead.  fcmf_bn_stats_blocks: those blocks -> the per-group totals inside `sums` (a fcmf_bn_stats_workspace
 * buffer) that fcmf_bn_finalize reads; rows_per_group must be a multiple of block_rows (else FCMF_ERR_UNSUPPORTED). */
int fcmf_gemm_colstats(fcmf_gemm_ctx* ctx, const void* A, const void* B, void* C, float* stats, int M, int N, int K, int64_t lda,
                       int64_t ldb, int64_t ldc, void* stream);
int fcmf_conv_gemm_colstats(fcmf_gemm_ctx* ctx, const void* x, const void* w, void* y, float* stats, int n, int Hp, int Wp, int C,
                            int Ho, int Wo, int kh, int kw, int stride, int Cout, void* stream);
int fcmf_gemm_colstats_block_rows(const fcmf_gemm_ctx* ctx, int M, int N, int K);
int fcmf_bn_stats_blocks(const float* blockstats, double* sums, int64_t rows_per_group, int groups, int C, int block_rows, void* stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on NHWC: [N,H,W,C] -> [N,(H-1)/2+1,(W-1)/2+1,C] */
int fcmf_maxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);
/* F.adaptive_avg_pool2d(x, [oh, ow]) of an NHWC activation, float32 output: layout 0 = [N, C, oh, ow] (what
 * myResNetImg returns, resnet_utils.py:24), layout 1 = [N, oh*ow, C] (the [B, 49, 2048] token layout the driver
 * builds with view/permute, run_multimodal_fcmf.py:451).  oh = ow = 1 is myResNetRoI's x.mean(3).mean(2) (:48). */
int fcmf_adaptive_avgpool(const void* x, float* y, int N, int H, int W, int C, int oh, int ow, int layout,
                          int dtype, void* stream);

/* ---- backward of the trunk (fine-tuning the CNN: resnet_utils.py if_fine_tune=True, --fine_tune_cnn) ----
 * BatchNorm2d (+ the ReLU behind it) backward, grouped like the forward.  g: gradient wrt the block output z; z: that
 * output (ReLU mask z > 0; NULL = no ReLU); y: the convolution output the forward normalised; mean/rstd from
 * fcmf_bn_finalize.  dy <- gradient wrt y: gamma*rstd*(gm - mean_rows(gm) - xhat*mean_rows(gm*xhat)) with gm = g*[z>0]
 * (training != 0) or gamma*rstd*gm (eval); gres (may be NULL) <- gm, the gradient of the identity branch that was added
 * before the ReLU; dgamma / dbeta float32 [C] are ACCUMULATED.  dy / gres may alias g.
 * sums: double workspace of fcmf_bn_stats_workspace(rows_per_group, groups, C) elements. */
int fcmf_bn_bwd(const void* g, const void* z, const void* y, const float* mean, const float* rstd, const float* gamma,
                double* sums, void* dy, void* gres, float* dgamma, float* dbeta, int64_t rows_per_group, int groups,
                int C, int training, int dtype, void* stream);
/* transpose of fcmf_conv_im2col (NHWC, same dtype both sides): dX[n,h,w,c] = sum of dA over the windows covering the
 * pixel, gathered per input pixel (no atomics).  dA [N*Ho*Wo, Kpad] = dY * W is the patch-matrix gradient. */
int fcmf_conv_col2im(const void* dA, void* dX, int N, int H, int W, int C, int kh, int kw, int stride, int pad,
                     int Kpad, int dtype, void* stream);
/* nn.MaxPool2d(3, 2, 1) backward: dX gets dY of every window whose first maximum (scan order) the pixel is */
int fcmf_maxpool3x3s2_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int dtype, void* stream);
/* F.adaptive_avg_pool2d backward: dy float32 in the forward's output layout (0 = [N,C,oh,ow], 1 = [N,oh*ow,C]) */
int fcmf_adaptive_avgpool_bwd(const float* dy, void* dx, int N, int H, int W, int C, int oh, int ow, int layout,
                              int dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Data-parallel gradient exchange (reference: torch.distributed.init_process_group("nccl") + DistributedDataParallel,
 * run_multimodal_fcmf.py:169,237-240; run_pretraining_fcmf.py:196-199).  One communicator per process (= per GPU, the HIP
 * device current at creation time); the flat gradient arena is all-reduced bucket by bucket IN PLACE on `stream`
 * (RCCL over xGMI), so the collective of one bucket overlaps the backward kernels that are still producing the next.
 * RCCL is bound at run time (the process's own copy if it has one, e.g. PyTorch's): FCMF_ERR_COMM when it is absent.
 *   fcmf_dp_unique_id   : rank 0 fills a HOST buffer of FCMF_DP_UNIQUE_ID_BYTES and hands it to the other ranks out of band
 *                         (the drivers use their torch.distributed store);
 *   fcmf_dp_comm_create : collective over the `nranks` processes holding that id; *comm receives an opaque handle;
 *   fcmf_dp_allreduce_bucket: buf[count] of `dtype` (FCMF_F32 / FCMF_BF16) <- sum (average = 0) or mean (average != 0, formed
 *                         inside the collective) over the ranks; enqueued on `stream`, returns at once. */
#define FCMF_DP_UNIQUE_ID_BYTES 128
int fcmf_dp_unique_id(void* out_host);
int fcmf_dp_comm_create(void** comm, const void* unique_id_host, int nranks, int rank);
int fcmf_dp_comm_destroy(void* comm);
int fcmf_dp_allreduce_bucket(void* comm, void* buf, int64_t count, int dtype, int average, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FCMF_HIP_H */
