/*
 * imt_hip.h -- C ABI of libimt_hip.so: the MI355X (gfx950) implementation of ImageTranslate's
 * transformer encoder-decoder train-step hot path.
 *
 * The reference (rasoolims/ImageTranslate) has NO native/FFI layer: its hot path is Python nn.Modules on
 * stock PyTorch + HuggingFace transformers==2.9.0.  Each entry point below therefore cites the reference
 * Python (or restated HF-BERT 2.9.0) computation it replaces; the Python classes of the same names in
 * imagetranslate_amd/ bind these through ctypes.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless named host_*.
 *   - `stream` is a hipStream_t passed as void*; the library never allocates/frees device memory and
 *     never synchronises the device; every result is complete in `stream` order when the call returns to
 *     the host.  One exception to "no global state": a whole-stack imt_stack_backward forks each layer's
 *     weight-gradient launch onto ONE library-owned non-blocking side stream (created on first use) and
 *     joins it back into `stream` with events before it returns -- so a stack backward is not capturable
 *     into a hipGraph on a single stream unless IMT_DW_SIDE_STREAM=0, and one process drives one GPU from
 *     one thread (the deployment model of this library).  The tuning aids IMT_TRACE / IMT_PROF do allocate
 *     and synchronise; they are off unless their environment variable is set.
 *   - return 0 on success, a negative IMT_ERR_* otherwise; imt_last_error() gives a thread-local message.
 *   - dtype: IMT_F32 (parity mode, exact fp32 MFMA/VALU) or IMT_BF16 (bf16 storage, fp32 accumulate).
 *     "T" below means the element type selected by `dtype`.  Gradients of PARAMETERS are always fp32
 *     and are ACCUMULATED (+=) into the caller's flat gradient buffer.
 *   - leading dimensions are in elements; rows must be 16-byte aligned (ld % 8 == 0 for bf16, % 4 for f32).
 */
#ifndef IMT_HIP_H
#define IMT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IMT_OK 0
#define IMT_ERR_BAD_ARG (-1)
#define IMT_ERR_UNSUPPORTED (-2)
#define IMT_ERR_LAUNCH (-3)

#define IMT_F32 0
#define IMT_BF16 1

int imt_version(void);
const char* imt_last_error(void);

/* Optional per-launch profiler used by bench.py for the roofline object: when enabled, every kernel launch is
 * bracketed by two hipEvents recorded on the launch stream.  imt_prof_report waits for the recorded events
 * (host-side), aggregates per kernel kind and clears the log.  flops/bytes are the ALGORITHMIC figures of the
 * launches (2*M*N*K per GEMM; minimal operand traffic for the HBM-bound kernels).  Off by default. */
/* sizeof() of the argument structures below as THIS library was compiled, by name ("imt_gemm_args", "imt_attn_args",
 * "imt_prof_row", "imt_attn_block", "imt_layer_desc", "imt_stack_desc", "imt_stack_io", "imt_attn_decode_args",
 * "imt_decode_io", "imt_beam_args", "imt_mass_args"); -1 for an unknown name.  A binding checks its own layout against it
 * once at load time (imagetranslate_amd/_lib.py does): a stale binding then fails loudly instead of passing shifted fields. */
int imt_abi_sizeof(const char* struct_name);

typedef struct imt_prof_row {
  char kind[48];
  int64_t launches;
  double total_ms;
  double flops;
  double bytes;
} imt_prof_row;
/* test utility: `blocks` workgroups of `threads` threads holding `lds_bytes` of LDS each spin for ~`cycles` shader clocks
 * (a stand-in for a long-running communication kernel when measuring the step under CU pressure). */
int imt_debug_spin(int blocks, int threads, int lds_bytes, int64_t cycles, void* stream);
int imt_prof_enable(int on);
int imt_prof_report(imt_prof_row* rows, int max_kinds);

/* ------------------------------------------------------------------ GEMM (all nn.Linear fwd/bwd on the path)
 * layout IMT_NT: C[M,N] = A[M,K] * B[N,K]^T   (y = x W^T : BertSelfAttention.query/key/value, *.dense,
 *                                              BertOutputLayer.layer -- src/bert_seq2seq.py:6-12)
 *        IMT_NN: C[M,N] = A[M,K] * B[K,N]     (dx = dy W)
 *        IMT_TN: C[M,N] = A[K,M]^T * B[K,N]   (dW = dy^T x)
 * epilogue, applied in this order to v = alpha * acc:
 *   bias     : v += bias[n]                                   (T)
 *   aux_mode : IMT_AUX_GELU_FWD  aux[m,n] = v ; v = gelu_erf(v)     (BertIntermediate, src/lm_config.py:7)
 *              IMT_AUX_DGELU     v *= gelu_erf'(aux[m,n])
 *   dropout  : v = keep(seed, m*N+n) ? v/(1-p) : 0            (hidden_dropout_prob, src/lm_config.py:8)
 *   resid    : v += resid[m,n]                                (T)   (BertSelfOutput / BertOutput residual)
 *   accumulate: v += C[m,n]
 *   store as c_dtype (T or IMT_F32).  split_k > 1: fp32 atomicAdd of the raw partial products into C
 *   (c_dtype must be IMT_F32; no other epilogue; C pre-initialised by the caller).
 */
#define IMT_NT 0
#define IMT_NN 1
#define IMT_TN 2
#define IMT_AUX_NONE 0
#define IMT_AUX_GELU_FWD 1
#define IMT_AUX_DGELU 2
#define IMT_AUX_SPLITK_WS 3 /* aux = fp32 WORKSPACE of split_k * M * N floats: NT / NN product as split_k K-ranges of 256 x 256-tile
                               workgroups, each into its own slab, then one reduce launch into C (C += with accumulate; plain
                               epilogue only: alpha / alpha_dev).  For few output tiles and a very long K (dX through the
                               vocabulary); fixed summation order, no atomics.  split_k = 2..16. */

typedef struct imt_gemm_args {
  int32_t dtype, layout;
  int32_t M, N, K;
  const void* A; int64_t lda;
  const void* B; int64_t ldb;
  void* C; int64_t ldc;
  int32_t c_dtype;
  int32_t accumulate;
  const void* bias;
  const void* resid; int64_t ldr;
  void* aux; int64_t ldaux;
  int32_t aux_mode;
  int32_t split_k;
  float alpha;
  float dropout_p;
  uint64_t dropout_seed;
  const float* alpha_dev; /* nullable device scalar multiplied into alpha (upstream loss gradient, no host sync) */
  float* a_colsum;        /* IMT_TN only, nullable: a_colsum[m] += alpha * sum_k A[k,m]  (bias gradient of the same
                             nn.Linear, fused into its weight-gradient GEMM: dy is read once) */
  int32_t force_general;  /* tests / tuning: kernel variant, 0 = auto, 1 = register-staged double buffer (2 blocks/CU),
                             2 / 4 = LDS-DMA 3- / 4-stage ring (1 block/CU; K a whole number of tiles), 3 = single LDS
                             buffer + register prefetch (3 blocks/CU), 5 = persistent wave-specialised (LDS-DMA
                             producer waves, K a whole number of tiles), 6 = 256 x 256 tiles (NT / NN, K a whole
                             number of tiles) */
  int32_t force_pipeline; /* reserved */
  /* LayerNorm of the finished rows of C, in the same launch (HF BertSelfOutput / BertOutput, src/bert_seq2seq.py:84-90,
   * 139-143: LayerNorm(dropout(dense(x)) + input) with the bias / dropout / residual epilogue above).  ln_out != NULL:
   * ln_out[M,N] (ld_ln, type c_dtype) = LayerNorm(C rows; ln_gamma, ln_beta, ln_eps), ln_mean / ln_rstd fp32 [M] (what
   * imt_layernorm_bwd takes).  C still receives the LayerNorm INPUT.  The default build runs imt_layernorm_fwd behind the
   * GEMM (C and ln_out contiguous).  A build with -DIMT_LN_TICKET=1 normalises in-launch where the persistent kernel runs
   * one tile per workgroup: C is stored write-through, each 128-row block has a ticket in ln_tickets (>= ceil(M / 128)
   * int32, ZERO on entry, zero again on return; may be NULL: second launch), and the workgroup whose column tile completes
   * a row block normalises it (no workgroup ever waits for another) -- measured slower than the second launch on MI355X
   * (DESIGN.md section 5). */
  const void* ln_gamma; const void* ln_beta;
  void* ln_out; int64_t ld_ln;
  float* ln_mean; float* ln_rstd;
  int32_t* ln_tickets;
  float ln_eps;
  int32_t reserved_ln;
  /* Few output tiles and a long K (the reference's inference and captioning callers: src/seq_gen.py:164-194 decodes
   * batch x beam = 320 rows per step, src/train_captioning.py:51-72 trains on 32 x 31 caption tokens; 128-row tiles then
   * occupy 12-52 of the 256 CUs and each walks the whole K alone).  With a caller-owned fp32 workspace here, imt_gemm may run
   * such a product (NT / NN, K >= 16 tiles) as 2..8 K-ranges per output tile on the persistent kernel, each range into its own
   * fp32 slab, followed by ONE launch that sums the slabs in a fixed order and applies the whole epilogue (bias, GELU /
   * GELU' with aux, dropout, residual, accumulate, C of either type): no atomics, bit-reproducible.  NULL / too small: never
   * taken.  imt_gemm_splitk_ws_bytes() is enough for every product whose 128 x 128 tiles times splits stay <= 256. */
  void* splitk_ws; int64_t splitk_ws_bytes;
} imt_gemm_args;
int64_t imt_gemm_splitk_ws_bytes(void);
int imt_gemm(const imt_gemm_args* a, void* stream);
/* All weight-gradient GEMMs (IMT_TN, fp32 C += alpha * A^T B, optional a_colsum) of one transformer layer in ONE
 * launch (HOST array of `count` descriptors).  Problems that cannot be grouped fall back to imt_gemm each. */
int imt_gemm_grouped_tn(const imt_gemm_args* list, int count, void* stream);

/* Dense + bias + dropout + residual + LayerNorm in ONE launch -- HF BertSelfOutput / BertOutput as instantiated by
 * src/bert_seq2seq.py:84-90,139-143: out = LayerNorm(dropout(x W^T + bias) + resid; eps), row-complete 32 x N tiles
 * (imagetranslate_amd/csrc/gemm_ln.hip).  x [M,K] (ldx), w [N,K] (ldw), bias / gamma / beta [N], resid [M,N] (ldr, may be
 * NULL), all of `dtype`; pre_ln [M,N] (may be NULL) receives the LayerNorm INPUT (what imt_layernorm_bwd reads as x), out
 * [M,N] its output (both with leading dimension ldo); mean / rstd fp32 [M] (may be NULL).  Dropout uses the element index
 * m * N + n like imt_gemm's epilogue, so imt_layernorm_bwd(dx_dropout_seed = dropout_seed) regenerates the mask.
 * Supported: N in {128, 256, 384, 512}, K a whole number of 128-byte tiles (imt_gemm_bias_residual_ln_supported). */
int imt_gemm_bias_residual_ln_supported(int dtype, int N, int K);
int imt_gemm_bias_residual_ln(int dtype, const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias,
                              const void* resid, int64_t ldr, const void* gamma, const void* beta, void* pre_ln, void* out,
                              int64_t ldo, float* mean, float* rstd, int M, int N, int K, float eps, float dropout_p,
                              uint64_t dropout_seed, void* stream);

/* column sums: out[n] += scale * sum_m X[m,n]  -> bias gradients (fp32, accumulated); scale_dev nullable. */
int imt_colsum(int dtype, const void* X, int64_t ldx, int M, int N, float* out, const float* scale_dev, void* stream);

/* ------------------------------------------------------------------ LayerNorm (torch.nn.LayerNorm, eps 1e-12)
 * fwd: y = (x - mean) * rstd * gamma + beta ; saves mean/rstd (fp32, [rows]) for backward.
 *      optional dropout on y (BertEmbeddings: dropout(LN(.))).
 * bwd: dx = LN'(dy) ; dgamma/dbeta accumulated (fp32 atomics) ; optional second output
 *      dx_drop = dropout-masked dx (grad w.r.t. the dense output that was dropped before the residual add).
 */
int imt_layernorm_fwd(int dtype, const void* x, const void* gamma, const void* beta, void* y, float* mean,
                      float* rstd, int rows, int d, float eps, float dropout_p, uint64_t dropout_seed,
                      void* stream);
/* y = dropout(LayerNorm(x + resid)), sum_out = x + resid (what imt_layernorm_bwd reads as its x): the residual add of
 * BertSelfOutput / BertOutput (src/bert_seq2seq.py:84-90) for callers whose dense layer did not fuse it. */
int imt_add_layernorm_fwd(int dtype, const void* x, const void* resid, const void* gamma, const void* beta, void* sum_out,
                          void* y, float* mean, float* rstd, int rows, int d, float eps, float dropout_p,
                          uint64_t dropout_seed, void* stream);
int imt_layernorm_bwd(int dtype, const void* dy, const void* x, const void* gamma, const float* mean,
                      const float* rstd, void* dx, float* dgamma, float* dbeta, int rows, int d,
                      float y_dropout_p, uint64_t y_dropout_seed, void* dx_drop, float dx_dropout_p,
                      uint64_t dx_dropout_seed, float* partial_ws, void* stream);
/* partial_ws: NULL, or an [IMT_LN_PARTIAL_COPIES][2][d] fp32 buffer ZEROED by the caller (IMT_LN_BWD_WS_FLOATS(d) floats).  When given, the
 * per-workgroup column sums are added to copy (workgroup % copies) of it INSTEAD of dgamma / dbeta: 256 workgroups adding
 * into the same d addresses serialise at the memory side (~6 us per launch at rows=8192, d=512), separate copies
 * (each XCD has its own) do not.  imt_ln_partial_reduce then folds any number of such buffers into the gradients in ONE launch:
 * grads[dgamma_off[i] + c] += sum_k partials[i][k][0][c], likewise dbeta (offsets are HOST arrays of element offsets
 * into `grads`, a negative dgamma offset skips that buffer; at most 224 buffers per call, laid out back to back).  (Also measured, slower: a same-launch "last
 * workgroup of a group sums its group" reduction -- its device-scope release fence writes back the XCD's L2.) */
#define IMT_LN_PARTIAL_COPIES 32   /* a multiple of the 8 XCDs: workgroup b -> copy b % 32 stays on XCD b % 8 */
#define IMT_LN_BWD_WS_FLOATS(d) (IMT_LN_PARTIAL_COPIES * 2 * (int64_t)(d))
int imt_ln_partial_reduce(const float* partials, int n_sites, int d, const int64_t* host_dgamma_off,
                          const int64_t* host_dbeta_off, float* grads, void* stream);

/* ------------------------------------------------------------------ embeddings (HF BertEmbeddings)
 * fwd: out[n,:] = word[ids[n]] + pos[pos_ids ? pos_ids[n] : n % seq_len] + type[type_ids[n]]   (pre-LN sum)
 * bwd: scatter-add d(sum) into the three fp32 gradient tables; word rows with id == pad_id get no gradient
 *      (nn.Embedding padding_idx).
 */
int imt_embed_fwd(int dtype, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                  const void* word, const void* pos, const void* type, void* out, int n_tokens, int seq_len,
                  int d, int vocab, int max_pos, int n_types, void* stream);
int imt_embed_bwd(int dtype, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                  const void* dsum, float* dword, float* dpos, float* dtype_tab, int n_tokens, int seq_len, int d,
                  int64_t pad_id, void* stream);
/* BertEmbeddings in ONE launch: sum_out[n,:] = word[ids[n]] + pos[...] + type[...] (kept for the backward),
 * y = dropout(LayerNorm(sum_out)) -- imt_embed_fwd + imt_layernorm_fwd without the round trip of the sum. */
int imt_embed_ln_fwd(int dtype, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids, const void* word,
                     const void* pos, const void* type, const void* gamma, const void* beta, void* sum_out, void* y,
                     float* mean, float* rstd, int n_tokens, int seq_len, int d, int vocab, int max_pos, int n_types, float eps,
                     float dropout_p, uint64_t dropout_seed, void* stream);

/* ------------------------------------------------------------------ attention (HF BertSelfAttention)
 * scores = Q K^T * scale + additive mask ; P = softmax ; [dropout(P)] ; O = P V, heads merged in the output.
 * Q/K/V are strided views: element (b, t, h, e) at base[(b*T + t)*ld + h*head_dim + e]
 * mask semantics == (1 - m) * -10000.0 added to the scaled scores (src/bert_seq2seq.py:25-38, HF
 * get_extended_attention_mask), m = AND of:
 *   key_mask[b, j]   (uint8, nullable)  -- encoder pad mask / encoder_attention_mask
 *   causal: j <= i   (flag)             -- future_mask / is_decoder 2-D mask
 *   query_mask[b, i] (uint8, nullable)  -- the tgt_mask.unsqueeze(-1) factor of future_mask (src/seq2seq.py:14-17)
 *   mask3d[b, i, j]  (uint8, nullable)  -- arbitrary 3-D tgt_attention_mask
 * lse[b,h,i] (fp32) = log-sum-exp of the masked scaled scores, saved for backward.
 * Backward kernels (bf16): Tq, Tk <= 128 one fused single-workgroup kernel per (batch, head); 129..256 (MASS, BASELINE
 * configs[4]: src/mass_seq2seq.py:32-60 runs the encoder over 256 tokens) its 256-key sibling; longer sequences and fp32 the
 * dQ + dK/dV kernel pair.  All are free of cross-workgroup sums: bit-reproducible.
 */
typedef struct imt_attn_args {
  int32_t dtype;
  int32_t B, H, Tq, Tk, head_dim;
  const void* Q; int64_t ldq;
  const void* K; int64_t ldk;
  const void* V; int64_t ldv;
  void* O; int64_t ldo;
  float* lse;
  const uint8_t* key_mask;
  const uint8_t* query_mask;
  const uint8_t* mask3d;
  int32_t causal;
  float scale;
  float dropout_p;
  uint64_t dropout_seed;
  /* backward only */
  const void* dO; int64_t lddo;
  void* dQ; int64_t lddq;
  void* dK; int64_t lddk;
  void* dV; int64_t lddv;
  float* delta; /* workspace [B,H,Tq] fp32: rowsum(dO * O) */
} imt_attn_args;
int imt_attention_fwd(const imt_attn_args* a, void* stream);
int imt_attention_bwd(const imt_attn_args* a, void* stream);
/* BertSelfAttention's query / key / value projections INSIDE the attention forward (round 3): one launch computes
 * q|k|v = x W^T + b for a short self-attention (bf16, head_dim 64, even number of heads, 64 < T <= 128, no 3-D mask; one
 * workgroup per batch element and pair of heads), stores them through a->Q / a->K / a->V -- which here are OUTPUT views of
 * the projection buffer, element (b, t, h, e) as above -- and continues with the attention of imt_attention_fwd on them:
 * a->O, a->lse as there.  x [B*T, d_model] (ldx), w [3 d_model, d_model] row-major with the query rows first, then key, then
 * value (the runtime's fused projection weight), bias [3 d_model] or NULL.  Bit-identical to imt_gemm + imt_attention_fwd. */
int imt_attention_qkv_fwd_supported(int dtype, int head_dim, int H, int Tq, int Tk, int d_model, int has_mask3d);
int imt_attention_qkv_fwd(const imt_attn_args* a, const void* x, int64_t ldx, const void* w, const void* bias, int d_model,
                          void* stream);

/* ------------------------------------------------------------------ row select (src/seq2seq.py:175-177)
 * gather: out[r,:] = x[idx[r],:] ; scatter (its backward): dx[idx[r],:] = dout[r,:] (dx pre-zeroed by caller
 * or zero_rest=1 to have the kernel write zeros to unselected rows given the inverse map).
 */
int imt_gather_rows(int dtype, const void* x, int64_t ldx, const int32_t* idx, void* out, int64_t ldo, int n_sel,
                    int d, void* stream);
int imt_scatter_rows(int dtype, const void* dout, int64_t ldo, const int32_t* idx, void* dx, int64_t ldx, int n_sel,
                     int d, void* stream);

/* ------------------------------------------------------------------ log-softmax + label-smoothed NLL
 * (F.log_softmax src/seq2seq.py:179-180 ; SmoothedNLLLoss src/loss.py:10-27 ; .mean() train_image_mt.py:282)
 * imt_log_softmax_fwd : lp = logits - logsumexp(logits) rowwise  (fp32 out, [N,V]); lse saved.
 * imt_smoothed_nll_fwd: loss[r] = (1-eps)*(-lp[r,t]) + (eps/V)*(-sum_v lp[r,v]) ; 0 where t == ignore_index.
 * imt_xent_fused_fwd_bwd: from raw logits (T, [N,V], in place): per-row loss and
 *      dlogits = grad_scale * (softmax - ((1-eps)*onehot + eps/V)), 0 on ignored rows, written over logits.
 */
int imt_log_softmax_fwd(int dtype, const void* logits, int64_t ld, float* lp, int64_t ldlp, float* lse, int N, int V,
                        void* stream);
int imt_log_softmax_bwd(const float* dlp, int64_t lddlp, const float* lp, int64_t ldlp, int out_dtype, void* dlogits,
                        int64_t ld, int N, int V, void* stream);
int imt_smoothed_nll_fwd(const float* lp, int64_t ldlp, const int64_t* target, float* loss, int N, int V,
                         float epsilon, int64_t ignore_index, void* stream);
int imt_smoothed_nll_bwd(const float* dloss, const int64_t* target, float* dlp, int64_t lddlp, int N, int V,
                         float epsilon, int64_t ignore_index, void* stream);
int imt_xent_fused_fwd_bwd(int dtype, void* logits, int64_t ld, const int64_t* target, float* loss_rows, int N,
                           int V, float epsilon, int64_t ignore_index, float grad_scale, void* stream);
/* out[0] = scale * sum(x[0..n)), fixed summation order: the `.mean()` of the per-row losses (train_image_mt.py:282). */
int imt_scaled_sum(const float* x, int n, float scale, float* out, void* stream);

/* ------------------------------------------------------------------ data-parallel gradient exchange (RCCL over xGMI)
 * Replaces torch DistributedDataParallel's NCCL all-reduce of src/train_image_mt.py:72-76 (bootstrap src/utils.py:93-97):
 * one process per GPU, one communicator per process.  Rank 0 calls imt_comm_get_unique_id and hands the
 * imt_comm_unique_id_bytes() (= 128) HOST bytes to every rank out of band; every rank then calls imt_comm_init with the
 * same id.  imt_comm_allreduce: in-place SUM of `count` elements over the ranks, asynchronous in `stream` (issue one call
 * per gradient bucket as the backward finalises it; fold 1 / world_size into imt_clip_adam's grad_scale);
 * imt_comm_broadcast: rank `root`'s buffer to all (the parameter broadcast of DDP's constructor).  librccl is opened at
 * first use (no load-time dependency); every function returns IMT_ERR_* with a message on RCCL errors. */
int imt_comm_unique_id_bytes(void);
int imt_comm_get_unique_id(void* host_id_out);
int imt_comm_init(const void* host_unique_id, int world_size, int rank, void** comm_out);
int imt_comm_allreduce(void* comm, void* buf, int64_t count, int dtype, void* stream);
int imt_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* stream);
int imt_comm_destroy(void* comm);

/* GEMM selection policy for data-parallel runs: share_cus != 0 makes the one-tile-per-CU products with K < 1024 take the
 * three-workgroups-per-CU kernel instead of the persistent one-workgroup-per-CU kernel, which needs a whole second round
 * as soon as a collective's resident kernels hold a few CUs.  Returns the previous setting.
 * The environment variable IMT_GEMM_SHARE_CUS (0 / 1), when set, overrides this call. */
int imt_set_gemm_share_cus(int share_cus);

/* ------------------------------------------------------------------ optimizer
 * (clip_grad_norm_ train_image_mt.py:291 ; AdamInverseSqrtWithWarmup src/utils.py:105-156)
 * imt_sumsq       : out[0] += sum(g^2) over n fp32 elements (out pre-zeroed by caller); DETERMINISTIC (fixed
 *                   reduction order: data-parallel replicas holding identical gradients get identical norms);
 *                   partial_ws = IMT_SUMSQ_WS_FLOATS floats of scratch.
 * imt_clip_adam   : clip_coef = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) ; g *= clip_coef ;
 *                   Adam(beta1, beta2, eps, no weight decay, bias-corrected) on fp32 master params ;
 *                   optionally writes the bf16 shadow copy ; optionally zeroes g.
 *                   lr and step are read from the host args (one launch per step).
 * imt_clip_scale  : g *= grad_scale * min(1, max_norm / (grad_scale * sqrt(sumsq[0]) + 1e-6)) in place: the clip of a
 *                   gradient-accumulation micro-step that does not end in an optimizer step (train_image_mt.py:291-295:
 *                   clip after every backward, step every `accum`).
 */
#define IMT_SUMSQ_WS_FLOATS 1024
int imt_sumsq(const float* g, int64_t n, float* out, float* partial_ws, void* stream);
int imt_clip_adam(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const float* sumsq,
                  float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps, int64_t step,
                  int zero_grad, void* stream);
int imt_clip_scale(float* g, int64_t n, const float* sumsq, float max_norm, float grad_scale, void* stream);
int imt_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int imt_gated_mix(int dtype, const void* a, const void* b, const void* gate, void* out, int64_t rows, int d,
                  void* stream);
/* Image head glue (ModifiedResnet head, src/image_model.py:37-41,77-78: dropout -> fc -> + location_embedding -> dropout;
 * the fc itself is imt_gemm): out[r, :] = dropout(x[r, :] + add[r % period, :]), x of in_dtype, out / add of out_dtype,
 * add may be NULL (plain dropout; with the forward's seed this is also the dropout's backward). */
int imt_add_rows_dropout(int in_dtype, const void* x, int out_dtype, void* out, const void* add, int64_t rows, int d,
                         int period, float dropout_p, uint64_t dropout_seed, void* stream);


/* ------------------------------------------------------------------ whole encoder / decoder stacks
 * The host-side runtime that chains the kernels above for BertEncoderModel.forward (src/bert_seq2seq.py:103-144)
 * and BertDecoderModel.forward (src/bert_seq2seq.py:40-91) and their backward passes.  All tensors live in
 * caller-owned memory: `params` is the FLAT parameter buffer in the compute dtype (fp32 master, or its bf16
 * shadow), `grads` the FLAT fp32 gradient buffer with the same element offsets; `ws` is a caller-allocated
 * workspace of imt_stack_workspace_bytes() bytes that carries the saved activations from forward to backward.
 * Offsets are in ELEMENTS; -1 = absent.  q|k|v weights (and biases) of one attention block are contiguous
 * ([3d,d]) so the three projections run as one GEMM.
 */
typedef struct imt_attn_block {
  int64_t qkv_w, qkv_b; /* self.query|key|value .weight [3d,d], .bias [3d] */
  int64_t o_w, o_b;     /* output.dense */
  int64_t ln_g, ln_b;   /* output.LayerNorm */
} imt_attn_block;

typedef struct imt_layer_desc {
  imt_attn_block self_attn;
  imt_attn_block cross_attn; /* qkv_w == -1 when the layer has no crossattention */
  int64_t ff1_w, ff1_b;      /* intermediate.dense [ff,d] */
  int64_t ff2_w, ff2_b;      /* output.dense [d,ff] */
  int64_t ln2_g, ln2_b;      /* output.LayerNorm */
  int64_t cross_kv_w, cross_kv_b; /* crossattention key|value weight [2d,d] / bias [2d] when they do NOT follow the query
                                     projection in the flat buffer (-1: they sit at cross_attn.qkv_w + d*d / qkv_b + d).
                                     When these blocks of ALL layers are contiguous in layer order ([L*2d, d], [L*2d]),
                                     the runtime projects the encoder states for every layer with ONE GEMM and forms
                                     d(encoder states) / dW of all layers with one GEMM each. */
} imt_layer_desc;

typedef struct imt_stack_desc {
  int32_t dtype;
  int32_t d, heads, ff, vocab, max_pos, n_types, n_layers;
  int32_t is_decoder;
  int32_t reserved;
  int64_t pad_id;
  float ln_eps, hidden_dropout, attn_dropout;
  float reserved_f;
  int64_t emb_word, emb_pos, emb_type, emb_ln_g, emb_ln_b;
  const imt_layer_desc* layers; /* HOST pointer, n_layers entries */
  const void* params;           /* device, compute dtype */
  float* grads;                 /* device, fp32 (may be NULL for inference) */
} imt_stack_desc;

typedef struct imt_stack_io {
  int32_t B, T;              /* batch, sequence length of this stack's own tokens */
  int32_t Tk;                /* decoder: encoder sequence length */
  int32_t training;          /* 1: apply dropout (seeded) */
  const int64_t* ids;        /* [B,T] */
  const int64_t* type_ids;   /* [B,T] or NULL (zeros) */
  const int64_t* pos_ids;    /* [B,T] or NULL (arange) */
  const uint8_t* key_mask;   /* self-attention key mask [B,T]: encoder attention_mask / decoder 2-D tgt mask, or NULL */
  const uint8_t* query_mask; /* decoder: tgt_mask factor of future_mask [B,T] or NULL */
  const uint8_t* mask3d;     /* decoder: arbitrary [B,T,T] tgt_attention_mask or NULL */
  int32_t causal;            /* decoder self-attention causal flag */
  int32_t reserved;
  const void* enc_states;    /* decoder: [B,Tk,d] (compute dtype) */
  const uint8_t* enc_mask;   /* decoder: encoder_attention_mask [B,Tk] or NULL (ones) */
  void* out;                 /* [B,T,d] final hidden states (compute dtype) */
  uint64_t dropout_seed;
  /* backward only */
  const void* d_out;         /* [B,T,d] gradient of `out` */
  void* d_enc_states;        /* decoder: [B,Tk,d] gradient w.r.t. enc_states, OVERWRITTEN */
  /* forward only, optional: HOST array of n_wait_events hipEvent_t (as void*, NULL entries allowed).  `stream` waits for
   * wait_events[0] before the embeddings are read and for wait_events[1 + l] before layer l: an optimizer step that is still
   * updating the parameters on another stream (segment by segment, in the order the forward needs them) then delays only
   * the layer it has not reached yet instead of the whole stack. */
  const void* const* wait_events;
  int32_t n_wait_events;
  int32_t reserved2;
} imt_stack_io;

int64_t imt_stack_workspace_bytes(const imt_stack_desc* m, int B, int T, int Tk);
int imt_stack_forward(const imt_stack_desc* m, const imt_stack_io* io, void* ws, int64_t ws_bytes, void* stream);
/* backward over layers [layer_lo, layer_hi) in reverse order (layer_hi == n_layers first); the embedding
 * backward runs when layer_lo == 0.  Splitting lets the caller launch gradient all-reduce buckets between
 * segments (RCCL on a side stream).  Parameter gradients are accumulated into m->grads.  A call that covers
 * the whole stack (layer_lo == 0, layer_hi == n_layers) runs the per-layer weight-gradient launches on the
 * library's side stream, concurrently with the next layer's input-gradient chain (joined before return). */
int imt_stack_backward(const imt_stack_desc* m, const imt_stack_io* io, void* ws, int64_t ws_bytes, int layer_lo,
                       int layer_hi, void* stream);

/* ------------------------------------------------------------------ incremental decoding + beam search
 * Replaces the per-step work of BeamDecoder.forward (src/seq_gen.py:131-227).  The reference re-runs the decoder
 * on the whole prefix every step (:164-166) and re-projects the encoder states to cross K/V in every layer of
 * every step; here each step processes ONE new position per hypothesis against
 *   - a self-attention cache  [n_layers][r_max][t_max][3d]  (q|k|v of every position, written in place by the
 *     fused QKV GEMM), addressed through a slot table so that beam re-ordering never copies the cache:
 *     slots[r, j] = cache row that holds position j of hypothesis r (its ancestor at the time j was decoded);
 *   - cross-attention K/V     [n_layers][B][Tk][2d]  projected once per sentence by imt_decode_begin and shared by
 *     the `rep` hypotheses of a sentence (row r reads sentence r / rep).
 * With the causal mask of BertDecoderModel (src/bert_seq2seq.py:69-71) the cached keys/values equal the
 * recomputed ones, so the last-position hidden state is the reference's `decoder_states[:, -1, :]` (:191).
 */
typedef struct imt_attn_decode_args {
  int32_t dtype;
  int32_t R, H, head_dim; /* hypotheses, heads, 32|64 */
  int32_t n_keys;         /* keys attended (self: pos+1; cross: Tk) */
  int32_t rep;            /* hypotheses per sentence (mask row and, without slots, K/V row = r / rep) */
  const void* Q; int64_t ldq;        /* query of hypothesis r at Q + r*ldq (+ h*head_dim) */
  const void* K; const void* V;      /* element (row, j) at base + row*ld_row + j*ld_pos (+ h*head_dim) */
  int64_t ld_row, ld_pos;
  const int32_t* slots; int64_t ld_slots; /* [R, >= n_keys] or NULL (row = r / rep) */
  const uint8_t* key_mask; int64_t ld_mask; /* [R/rep, n_keys]: (1-m)*-10000 added to the scaled score; nullable */
  void* O; int64_t ldo;
  float scale;
  int32_t reserved;
} imt_attn_decode_args;
int imt_attention_decode(const imt_attn_decode_args* a, void* stream);

typedef struct imt_decode_io {
  int32_t R;               /* hypothesis rows this step (B at the first step, B*beam afterwards) */
  int32_t rep;             /* rows per source sentence */
  int32_t pos;             /* 0-based position being decoded == number of cached positions */
  int32_t Tk;              /* encoder length */
  int32_t t_max, r_max;    /* cache capacity: positions, rows */
  const int64_t* ids;      /* [R] newest token of every hypothesis */
  const int64_t* type_ids; /* [R] or NULL (zeros) */
  const int64_t* pos_ids;  /* [R] position-embedding index (normally == pos) */
  const int32_t* slots;    /* [R, t_max] slot table, slots[r, pos] == r ; NULL: every row reads its own cache row */
  const uint8_t* enc_mask; /* [R/rep, Tk] encoder_attention_mask or NULL (ones) */
  void* self_cache;        /* imt_decode_self_cache_bytes() */
  const void* cross_kv;    /* imt_decode_cross_bytes(), filled by imt_decode_begin */
  void* out;               /* [R, d] last-position hidden states (compute dtype) */
} imt_decode_io;
int64_t imt_decode_workspace_bytes(const imt_stack_desc* m, int r_max);
int64_t imt_decode_self_cache_bytes(const imt_stack_desc* m, int r_max, int t_max);
int64_t imt_decode_cross_bytes(const imt_stack_desc* m, int B, int Tk);
/* cross K|V of every decoder layer from the encoder states [B,Tk,d] (one GEMM per layer). */
int imt_decode_begin(const imt_stack_desc* m, const void* enc_states, int B, int Tk, void* cross_kv, void* stream);
int imt_decode_step(const imt_stack_desc* m, const imt_decode_io* io, void* ws, int64_t ws_bytes, void* stream);
/* bf16 stacks with hidden size 512 or 768 in heads of 64 run a step as ONE launch whose workgroups meet at grid-wide barriers with
 * bounded waits (csrc/decode_fused.hip; IMT_DECODE_FUSED=0: the launch-per-operator chain).  A wait that runs out abandons
 * the launch and sets a status word in `ws` (cleared by the step with pos == 0).  imt_decode_check synchronises `stream`
 * and returns IMT_ERR_LAUNCH if any step since then was abandoned, IMT_OK otherwise (and always for other stacks):
 * call it once when a search ends, before trusting its tokens. */
int imt_decode_check(const imt_stack_desc* m, int r_max, const void* ws, void* stream);

/* One beam-search step on device (src/seq_gen.py:193-227): log-softmax of the [B*rep, V] logits, the EOS /
 * length-limit zeroing (:194-196), length-penalised scores (:197-200, pow((len+6)/6, ratio)), top-`beam` over the
 * rep*V continuations of each sentence with ties broken by the LOWEST flat index (torch.topk leaves it unspecified),
 * the reference's PAD overwrites (:205-212, including that finished/over-limit slots take beam 0 as parent because
 * the overwritten flat index is divided by V, :216 with floor division), and the bookkeeping (:214-227): token
 * history, hypothesis sizes, scores, EOS flags, and the slot table of the self-attention cache.
 * Row r of the inputs is hypothesis (r / rep, r % rep); outputs have B*beam rows.  `step` is the reference's loop
 * index i (>= 1): inputs hold i tokens, outputs i+1.  eos_count[step] += number of output hypotheses containing EOS.
 */
typedef struct imt_beam_args {
  int32_t B, beam, rep, V;
  int32_t step, t_max;
  const float* logits; int64_t ld; /* [B*rep, V] fp32 */
  const float* scores_in;          /* [B*rep] */
  const float* sizes_in;           /* [B*rep] (ignored when beam == 1) */
  const uint8_t* eos_in;           /* [B*rep] hypothesis already contains EOS */
  const int64_t* max_lens;         /* [B] */
  const int64_t* hist_in;          /* [B*rep, t_max] */
  const int32_t* slots_in;         /* [B*rep, t_max] or NULL */
  float len_penalty_ratio;
  int32_t reserved;
  int64_t pad_idx, eos;
  float* cand_scores; int32_t* cand_idx; /* workspace [B*rep, beam] each */
  float* scores_out; float* sizes_out; uint8_t* eos_out; /* [B*beam] */
  int64_t* hist_out;               /* [B*beam, t_max] */
  int32_t* slots_out;              /* [B*beam, t_max] or NULL */
  int32_t* parent_out;             /* [B*beam] input row each output hypothesis extends */
  int64_t* tokens_out;             /* [B*beam] appended token */
  int32_t* eos_count;              /* [t_max] (pre-zeroed by the caller) or NULL */
} imt_beam_args;
int imt_beam_step(const imt_beam_args* a, void* stream);

/* ------------------------------------------------------------------ MASS batch construction on device
 * mass_mask / mass_unmask of src/utils.py:41-82.  Per sentence (row): a contiguous span of int(pad_index/2) tokens
 * starting at `first` is hidden from the encoder, where first = 1 (20%), the bound ceil(pad - (1-p)*pad) (20%) or
 * uniform in [2, bound] (60%) (src/utils.py:52-60); the decoder input `to_recover` is the span shifted right by one
 * with its ORIGINAL positions; hidden tokens become <mask> (80%), a random non-special id (10%) or stay (10%).
 * The random draws are counter-based on (seed, stream, index) -- NOT Python's generator: the same statistical
 * procedure, reproducible on device.  Sizes that depend only on pad_indices are computed by the caller:
 *   row_offsets[r] = sum_{q<r} int(pad_indices[q]/2)   (exclusive prefix sum; total = number of targets)
 *   recover_width  = max_r int(pad_indices[r]/2) + 1
 * src_text is modified in place (like the reference); imt_mass_unmask restores it from `targets`.
 */
typedef struct imt_mass_args {
  int32_t n_rows, width, recover_width;
  int32_t n_special, vocab;  /* replacement ids are drawn from [n_special, vocab) */
  float mask_prob;
  uint64_t seed;
  int64_t mask_id, pad_id;
  int64_t* src_text;            /* [n_rows, width] in/out */
  const int64_t* pad_indices;   /* [n_rows] index of the first pad token (width-1 if none) */
  const int64_t* row_offsets;   /* [n_rows] */
  uint8_t* src_mask;            /* [n_rows, width] out: 1 on hidden positions */
  int64_t* to_recover;          /* [n_rows, recover_width] out, padded with pad_id */
  int64_t* positions;           /* [n_rows, recover_width] out, padded with width-1 */
  int64_t* targets;             /* [total] out: hidden tokens in row-major order (== the reference's mask_idx) */
} imt_mass_args;
int imt_mass_mask(const imt_mass_args* a, void* stream);
int imt_mass_unmask(int64_t* src_text, const uint8_t* src_mask, const int64_t* originals, const int64_t* row_offsets,
                    int n_rows, int width, void* stream);

/* Non-pad target selection (src/seq2seq.py:175-177 `flat[tgt_mask[:, 1:]]`, train_image_mt.py:253-256 targets):
 * idx[k] = flat position b*T1 + t of the k-th set mask[b, col0 + t] (row-major order), targets[k] = ids[b, col0 + t],
 * count[0] = number selected.  idx / targets need room for B*T1 entries. */
int imt_select_plan(const uint8_t* mask, int64_t ld_mask, const int64_t* ids, int64_t ld_ids, int B, int T1, int col0,
                    int32_t* idx, int64_t* targets, int32_t* count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IMT_HIP_H */
