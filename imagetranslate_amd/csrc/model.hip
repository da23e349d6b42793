// Host-side runtime: chains the kernels for a whole BERT encoder / cross-attending decoder stack
// (BertEncoderModel.forward src/bert_seq2seq.py:103-144, BertDecoderModel.forward :40-91, with the HF-BERT
// 2.9.0 layer semantics restated in SURVEY section 8: post-LN, erf-GELU, additive -10000 masks) and their
// backward passes.  Pure enqueue: no allocation, no synchronisation; every launch goes to the caller's stream
// so the whole forward/backward is capturable in a hipGraph.  Activations needed by backward are kept in the
// caller-owned workspace, carved identically by forward and backward.
#include "common.hpp"
#include "decode_fused.hpp"

namespace {

struct Carver {
  char* base; int64_t off;
  explicit Carver(void* b) : base(reinterpret_cast<char*>(b)), off(0) {}
  void* take(int64_t bytes) {
    void* p = base ? base + off : nullptr;
    off += (bytes + 255) & ~(int64_t)255;
    return p;
  }
};

struct AttnWs {  // one attention block (self or cross)
  void* qkv;     // self: [N,3d] ; cross: q [N,d] then kv [Nk,2d]
  void* kv;      // cross only
  int64_t kv_ld; // leading dimension of kv (2d, or L*2d when the projections of all layers are one buffer)
  void* ctx;     // [N,d]
  float* lse;    // [B,H,T]
  void* pre_ln;  // [N,d]  dense + residual (LN input)
  float* mean; float* rstd;
  void* out;     // [N,d]  LN output
};
struct LayerWs {
  AttnWs self_attn, cross;
  void* z; void* h;  // [N,ff]
  void* pre_ln2; float* mean2; float* rstd2;
  void* out;         // [N,d]
};
struct StackWs {
  void* emb_sum; float* emb_mean; float* emb_rstd; void* x0;
  LayerWs* layers;  // host array (carved on the stack of the caller)
  // backward scratch.  Every dy operand of a weight-gradient GEMM keeps its own buffer until the end of the layer,
  // because all dW GEMMs of a layer are deferred into ONE grouped launch (imt_gemm_grouped_tn).
  void* d_run;                       // [N,d]  running gradient w.r.t. the current block output
  void* dpre[3]; void* ddrop[3];     // [N,d]  LN-backward outputs of the FFN / cross / self blocks (residual / dense path)
  void* d_ctx;                       // [N,d]
  void* d_ff;                        // [N,ff]
  void* d_qkv;                       // [N,3d]
  void* d_q;                         // [N,d]   cross-attention query gradient
  void* d_kv;                        // [Nk,2d]
  void* d_emb;                       // [N,d]  gradient w.r.t. the embedding sum (its own buffer: see below)
  // The seven kinds of buffer above exist TWICE (even / odd layers): the weight-gradient launch of layer l may run on
  // the side stream while layer l-1 already writes its own dy operands (imt_stack_backward); use_scratch(l) points the
  // fields above at the set of layer l.
  struct Scratch { void* dpre[3]; void* ddrop[3]; void* d_ctx; void* d_ff; void* d_qkv; void* d_q; void* d_kv; } scr[2];
  void use_scratch(int l) {
    const Scratch& z = scr[l & 1];
    for (int i = 0; i < 3; ++i) { dpre[i] = z.dpre[i]; ddrop[i] = z.ddrop[i]; }
    d_ctx = z.d_ctx; d_ff = z.d_ff; d_qkv = z.d_qkv; d_q = z.d_q; d_kv = z.d_kv;
  }
  void* kv_all; void* d_kv_all;      // [Nk, L*2d] when the cross-attention key|value projections are batched
  float* delta;                      // [B,H,T]
  float* ln_partial;                 // [3 * layers + 1][IMT_LN_BWD_WS_FLOATS(d)]: per-XCD dgamma|dbeta partial sums of every
                                     // LayerNorm site (layer l: 3l = FFN, 3l+1 = cross, 3l+2 = self; last = embeddings)
  float* ln_site(int site, int d) const { return ln_partial + (int64_t)site * IMT_LN_BWD_WS_FLOATS(d); }
  int32_t* ln_tickets;               // [ceil(N / 128)] row-block tickets of imt_gemm's in-launch LayerNorm (zero between launches)
  int64_t ln_ticket_bytes;
  void* splitk; int64_t splitk_bytes; // fp32 slabs of imt_gemm's split-K mode (few output tiles: captioning / short batches); one
                                     // product at a time on the main stream, the grouped weight gradients never use it
  int64_t bytes;
};

constexpr int MAX_LAYERS = 64;

int64_t esize(int dtype) { return dtype == IMT_BF16 ? 2 : 4; }

// crossattention key|value weight / bias offsets of a layer (imt_layer_desc.cross_kv_*)
int64_t kv_w_off(const imt_stack_desc* m, const imt_layer_desc& p) { return p.cross_kv_w >= 0 ? p.cross_kv_w : p.cross_attn.qkv_w + (int64_t)m->d * m->d; }
int64_t kv_b_off(const imt_stack_desc* m, const imt_layer_desc& p) { return p.cross_kv_b >= 0 ? p.cross_kv_b : p.cross_attn.qkv_b + m->d; }
// all layers' key|value projections contiguous in layer order -> one [L*2d, d] weight: ONE GEMM projects the encoder
// states for every layer (forward), one forms d(encoder states) and one the weight gradients (backward)
bool cross_kv_batched(const imt_stack_desc* m) {
  if (!m->is_decoder || m->n_layers < 2) return false;
  const int64_t d = m->d;
  for (int l = 0; l < m->n_layers; ++l) {
    const imt_layer_desc& p = m->layers[l];
    if (p.cross_attn.qkv_w < 0 || p.cross_kv_w < 0 || p.cross_kv_b < 0) return false;
    if (p.cross_kv_w != m->layers[0].cross_kv_w + (int64_t)l * 2 * d * d) return false;
    if (p.cross_kv_b != m->layers[0].cross_kv_b + (int64_t)l * 2 * d) return false;
  }
  return true;
}

void carve(const imt_stack_desc* m, int B, int T, int Tk, void* ws, StackWs& w, LayerWs* layers) {
  Carver c(ws);
  const int64_t N = (int64_t)B * T, Nk = (int64_t)B * Tk, d = m->d, ff = m->ff, es = esize(m->dtype);
  w.emb_sum = c.take(N * d * es); w.emb_mean = (float*)c.take(N * 4); w.emb_rstd = (float*)c.take(N * 4);
  w.x0 = c.take(N * d * es);
  w.layers = layers;
  const bool batched = cross_kv_batched(m);
  const int64_t L2d = (int64_t)m->n_layers * 2 * d;
  w.kv_all = w.d_kv_all = nullptr;
  if (batched) { w.kv_all = c.take(Nk * L2d * es); w.d_kv_all = c.take(Nk * L2d * es); }
  for (int l = 0; l < m->n_layers; ++l) {
    LayerWs& L = layers[l];
    AttnWs& s = L.self_attn;
    s.qkv = c.take(N * 3 * d * es); s.kv = nullptr; s.ctx = c.take(N * d * es);
    s.lse = (float*)c.take((int64_t)B * m->heads * T * 4);
    s.pre_ln = c.take(N * d * es); s.mean = (float*)c.take(N * 4); s.rstd = (float*)c.take(N * 4);
    s.out = c.take(N * d * es);
    AttnWs& x = L.cross;
    memset(&x, 0, sizeof(x));
    if (m->is_decoder && m->layers[l].cross_attn.qkv_w >= 0) {
      x.qkv = c.take(N * d * es); x.ctx = c.take(N * d * es);
      if (batched) { x.kv = w.kv_all ? (void*)((char*)w.kv_all + (int64_t)l * 2 * d * es) : nullptr; x.kv_ld = L2d; }
      else { x.kv = c.take(Nk * 2 * d * es); x.kv_ld = 2 * d; }
      x.lse = (float*)c.take((int64_t)B * m->heads * T * 4);
      x.pre_ln = c.take(N * d * es); x.mean = (float*)c.take(N * 4); x.rstd = (float*)c.take(N * 4);
      x.out = c.take(N * d * es);
    }
    L.z = c.take(N * ff * es); L.h = c.take(N * ff * es);
    L.pre_ln2 = c.take(N * d * es); L.mean2 = (float*)c.take(N * 4); L.rstd2 = (float*)c.take(N * 4);
    L.out = c.take(N * d * es);
  }
  w.d_run = c.take(N * d * es);
  for (int k = 0; k < 2; ++k) {
    StackWs::Scratch& z = w.scr[k];
    for (int i = 0; i < 3; ++i) { z.dpre[i] = c.take(N * d * es); z.ddrop[i] = c.take(N * d * es); }
    z.d_ctx = c.take(N * d * es);
    z.d_ff = c.take(N * ff * es);
    z.d_qkv = c.take(N * 3 * d * es);
    z.d_q = c.take(N * d * es);
    z.d_kv = c.take((Nk > 0 ? Nk : 1) * 2 * d * es);
  }
  w.use_scratch(0);
  w.d_emb = c.take(N * d * es);
  w.delta = (float*)c.take((int64_t)B * m->heads * T * 4);
  w.ln_partial = (float*)c.take((int64_t)(3 * m->n_layers + 1) * IMT_LN_BWD_WS_FLOATS(d) * 4);
  w.ln_ticket_bytes = ((N + 127) / 128 + 1) * 4;
  w.ln_tickets = (int32_t*)c.take(w.ln_ticket_bytes);
  // split-K slabs: only batches whose products have <= 128 output tiles can take that path (imt_gemm: splitk_choice)
  w.splitk_bytes = ((N + 127) / 128) * ((d + 127) / 128) <= 128 ? imt_gemm_splitk_ws_bytes() : 0;
  w.splitk = w.splitk_bytes ? c.take(w.splitk_bytes) : nullptr;
  w.bytes = c.off;
}

struct Ctx {
  const imt_stack_desc* m; hipStream_t st; int dtype; int64_t es;
  int32_t* ln_tickets = nullptr;  // zeroed row-block tickets (imt_stack_forward); null: dense + LayerNorm as two launches
  void* splitk = nullptr; int64_t splitk_bytes = 0;  // imt_gemm_args.splitk_ws of every main-stream product
  const char* P(int64_t off) const { return reinterpret_cast<const char*>(m->params) + off * es; }
  float* G(int64_t off) const { return m->grads + off; }
};

#define RC(x) do { int rc__ = (x); if (rc__ != IMT_OK) return rc__; } while (0)

// y[M,N] = epilogue(x[M,K] W[N,K]^T)
int linear_fwd(const Ctx& c, const void* x, int64_t ldx, int M, int K, int64_t w_off, int64_t b_off, int N, void* y, int64_t ldy,
               const void* resid, int64_t ldr, void* aux, int aux_mode, float drop_p, uint64_t seed) {
  imt_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.dtype = c.dtype; a.layout = IMT_NT; a.M = M; a.N = N; a.K = K;
  a.A = x; a.lda = ldx; a.B = c.P(w_off); a.ldb = K; a.C = y; a.ldc = ldy; a.c_dtype = c.dtype;
  a.bias = b_off >= 0 ? c.P(b_off) : nullptr;
  a.resid = resid; a.ldr = ldr; a.aux = aux; a.ldaux = N; a.aux_mode = aux_mode; a.split_k = 1; a.alpha = 1.f;
  a.dropout_p = drop_p; a.dropout_seed = seed;
  a.splitk_ws = c.splitk; a.splitk_ws_bytes = c.splitk_bytes;
  return imt_gemm(&a, c.st);
}

// dx[M,K] = epilogue(dy[M,N] W[N,K])        (W stored [N,K] row-major -> NN with B = W)
int linear_bwd_input(const Ctx& c, const void* dy, int64_t lddy, int M, int N, int64_t w_off, int K, void* dx, int64_t lddx,
                     const void* resid, int64_t ldr, void* aux, int aux_mode, int accumulate) {
  imt_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.dtype = c.dtype; a.layout = IMT_NN; a.M = M; a.N = K; a.K = N;
  a.A = dy; a.lda = lddy; a.B = c.P(w_off); a.ldb = K; a.C = dx; a.ldc = lddx; a.c_dtype = c.dtype;
  a.resid = resid; a.ldr = ldr; a.aux = aux; a.ldaux = K; a.aux_mode = aux_mode; a.split_k = 1; a.alpha = 1.f;
  a.accumulate = accumulate;
  a.splitk_ws = c.splitk; a.splitk_ws_bytes = c.splitk_bytes;
  return imt_gemm(&a, c.st);
}

// y = LayerNorm(dropout(x W^T + b) + resid)  (BertSelfOutput / BertOutput): imt_gemm with the residual epilogue +
// imt_layernorm_fwd, or ONE launch of the row-complete fused kernel (gemm_ln.hip).  Measured on MI355X
// (profiles/r02_gemm_ln_study.txt): the fused kernel streams the whole weight through every workgroup (32-row tiles), which
// at d = 512 costs what the LayerNorm launch saves (8192 x 512 x 512: 22.2 us against 13.8 + 8.1, C1 step 7.63 against 7.61
// ms; incremental decoding 0.92 against 0.77 ms per step: its K loop is one serial stream per workgroup) -- it wins only for
// narrow models (N <= 256: 0.5-0.7x of the pair).  IMT_GEMM_LN=0 / 1: never / whenever supported.
bool fuse_dense_ln(const Ctx& c, int M, int N, int K) {
  static const int mode = getenv("IMT_GEMM_LN") ? atoi(getenv("IMT_GEMM_LN")) : -1;
  if (mode == 0 || !imt_gemm_bias_residual_ln_supported(c.dtype, N, K)) return false;
  if (mode == 1) return true;
  return N <= 256 && K <= 1024;
}
int dense_resid_ln(const Ctx& c, const void* x, int64_t ldx, int M, int K, int64_t w_off, int64_t b_off, int N, const void* resid,
                   int64_t g_off, int64_t beta_off, void* pre_ln, void* out, float* mean, float* rstd, float drop_p, uint64_t seed) {
  if (fuse_dense_ln(c, M, N, K))
    return imt_gemm_bias_residual_ln(c.dtype, x, ldx, c.P(w_off), K, b_off >= 0 ? c.P(b_off) : nullptr, resid, N, c.P(g_off), c.P(beta_off),
                                     pre_ln, out, N, mean, rstd, M, N, K, c.m->ln_eps, drop_p, seed, c.st);
  // imt_gemm with the LayerNorm of the finished rows: in the GEMM's own launch where it is a one-tile-per-workgroup launch
  // of the persistent kernel (the C1 shapes: the last column tile of a 128-row block normalises it), else a second launch
  imt_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.dtype = c.dtype; a.layout = IMT_NT; a.M = M; a.N = N; a.K = K;
  a.A = x; a.lda = ldx; a.B = c.P(w_off); a.ldb = K; a.C = pre_ln; a.ldc = N; a.c_dtype = c.dtype;
  a.bias = b_off >= 0 ? c.P(b_off) : nullptr;
  a.resid = resid; a.ldr = N; a.aux_mode = IMT_AUX_NONE; a.split_k = 1; a.alpha = 1.f;
  a.dropout_p = drop_p; a.dropout_seed = seed;
  a.ln_gamma = c.P(g_off); a.ln_beta = c.P(beta_off); a.ln_out = out; a.ld_ln = N; a.ln_mean = mean; a.ln_rstd = rstd;
  a.ln_eps = c.m->ln_eps; a.ln_tickets = c.ln_tickets;
  a.splitk_ws = c.splitk; a.splitk_ws_bytes = c.splitk_bytes;
  return imt_gemm(&a, c.st);
}

// Weight-gradient GEMMs of one layer are collected here and issued as ONE grouped launch at the end of the layer's
// backward (dW[N,K] += dy[M,N]^T x[M,K] ; db[N] += colsum(dy), fused).
struct DeferredDW {
  imt_gemm_args list[8];
  int n = 0;
  void add(const Ctx& c, const void* dy, int64_t lddy, const void* x, int64_t ldx, int M, int N, int K, int64_t w_off, int64_t b_off) {
    imt_gemm_args& a = list[n++];
    memset(&a, 0, sizeof(a));
    a.dtype = c.dtype; a.layout = IMT_TN; a.M = N; a.N = K; a.K = M;
    a.A = dy; a.lda = lddy; a.B = x; a.ldb = ldx; a.C = c.G(w_off); a.ldc = K; a.c_dtype = IMT_F32; a.alpha = 1.f;
    a.split_k = 1; a.accumulate = 1;
    a.a_colsum = b_off >= 0 ? c.G(b_off) : nullptr;  // dy is read once for dW and db
    // used only if the group cannot be formed (ragged token count): about one workgroup per CU via split-K
    const int tiles = imt_cdiv(N, 128) * imt_cdiv(K, 128);
    int sk = 1;
    while (sk < 8 && tiles * sk * 2 <= 256 && M / (sk * 2) >= 512) sk *= 2;
    if (M % (c.dtype == IMT_BF16 ? 64 : 32) != 0) { a.split_k = sk; a.accumulate = (sk == 1); }
  }
  int flush(hipStream_t st) {
    const int rc = n ? imt_gemm_grouped_tn(list, n, st) : IMT_OK;
    n = 0;
    return rc;
  }
};

// Side stream for the weight-gradient launches (one process drives one GPU from one thread: plain statics).
// dW of layer l only has to be complete when the stack's backward returns, and it is bound by operand traffic, while
// the dX chain of layer l-1 is a string of latency-bound launches: on two streams the hardware runs workgroups of
// both.  ready[l]: main stream has produced every dy operand of layer l.  done[l]: the side stream has finished dW(l).
struct SideQueue {
  hipStream_t st = nullptr;
  hipEvent_t ready[MAX_LAYERS], done[MAX_LAYERS];
  bool ok = false;
  bool init() {
    if (st) return ok;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { st = (hipStream_t)1; return ok = false; }
    for (int i = 0; i < MAX_LAYERS; ++i)
      if (hipEventCreateWithFlags(&ready[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) return ok = false;
    return ok = true;
  }
};
SideQueue g_side;

uint64_t site_seed(uint64_t base, int layer, int site) { return base + 0x9E3779B97F4A7C15ull * (uint64_t)(layer * 16 + site + 1); }

struct MaskSet { const uint8_t* key; const uint8_t* query; const uint8_t* m3d; int causal; };

int attn_args(const Ctx& c, imt_attn_args& a, int B, int Tq, int Tk, const void* q, int64_t ldq, const void* k, int64_t ldk,
              const void* v, int64_t ldv, void* o, float* lse, const MaskSet& ms, float drop_p, uint64_t seed) {
  memset(&a, 0, sizeof(a));
  a.dtype = c.dtype; a.B = B; a.H = c.m->heads; a.Tq = Tq; a.Tk = Tk; a.head_dim = c.m->d / c.m->heads;
  a.Q = q; a.ldq = ldq; a.K = k; a.ldk = ldk; a.V = v; a.ldv = ldv; a.O = o; a.ldo = c.m->d; a.lse = lse;
  a.key_mask = ms.key; a.query_mask = ms.query; a.mask3d = ms.m3d; a.causal = ms.causal;
  a.scale = 1.0f / sqrtf((float)a.head_dim);
  a.dropout_p = drop_p; a.dropout_seed = seed;
  return IMT_OK;
}

inline const char* offp(const void* p, int64_t elems, int64_t es) { return reinterpret_cast<const char*>(p) + elems * es; }
inline char* offp(void* p, int64_t elems, int64_t es) { return reinterpret_cast<char*>(p) + elems * es; }

// ------------------------------------------------------------------------------------------------ forward
// attention block (BertAttention = BertSelfAttention + BertSelfOutput); kv_src == nullptr -> self attention
int attn_block_fwd(const Ctx& c, const imt_attn_block& p, AttnWs& w, const void* x, int B, int T, const void* kv_src, int Tk,
                   const MaskSet& ms, bool training, uint64_t seed, int layer, int site0, int64_t kv_w = -1, int64_t kv_b = -1,
                   bool kv_ready = false) {
  const int d = c.m->d;
  const int N = B * T;
  const float hp = training ? c.m->hidden_dropout : 0.f, ap = training ? c.m->attn_dropout : 0.f;
  imt_attn_args a;
  if (!kv_src) {
    attn_args(c, a, B, T, T, w.qkv, 3 * d, offp(w.qkv, d, c.es), 3 * d, offp(w.qkv, 2 * d, c.es), 3 * d, w.ctx, w.lse, ms, ap,
              site_seed(seed, layer, site0));
    // short self-attention in bf16: the q|k|v projection runs inside the attention launch (bit-identical to the pair)
    static const bool qkv_fusion = getenv("IMT_ATTN_QKV_FUSION") && atoi(getenv("IMT_ATTN_QKV_FUSION")) != 0;  // measured neutral on C1: opt-in
    if (qkv_fusion && p.qkv_b >= 0 && imt_attention_qkv_fwd_supported(c.dtype, a.head_dim, a.H, T, T, d, ms.m3d != nullptr) &&
        (int64_t)B * T * d * c.es < (1ll << 31)) {
      RC(imt_attention_qkv_fwd(&a, x, d, c.P(p.qkv_w), c.P(p.qkv_b), d, c.st));
      RC(dense_resid_ln(c, w.ctx, d, N, d, p.o_w, p.o_b, d, x, p.ln_g, p.ln_b, w.pre_ln, w.out, w.mean, w.rstd, hp, site_seed(seed, layer, site0 + 1)));
      return IMT_OK;
    }
    RC(linear_fwd(c, x, d, N, d, p.qkv_w, p.qkv_b, 3 * d, w.qkv, 3 * d, nullptr, 0, nullptr, IMT_AUX_NONE, 0.f, 0));
  } else {
    const int Nk = B * Tk;
    RC(linear_fwd(c, x, d, N, d, p.qkv_w, p.qkv_b, d, w.qkv, d, nullptr, 0, nullptr, IMT_AUX_NONE, 0.f, 0));
    if (!kv_ready)  // (batched: the key|value projections of all layers were formed by one GEMM before the layer loop)
      RC(linear_fwd(c, kv_src, d, Nk, d, kv_w, kv_b, 2 * d, w.kv, w.kv_ld, nullptr, 0, nullptr, IMT_AUX_NONE, 0.f, 0));
    attn_args(c, a, B, T, Tk, w.qkv, d, w.kv, w.kv_ld, offp(w.kv, d, c.es), w.kv_ld, w.ctx, w.lse, ms, ap, site_seed(seed, layer, site0));
  }
  RC(imt_attention_fwd(&a, c.st));
  RC(dense_resid_ln(c, w.ctx, d, N, d, p.o_w, p.o_b, d, x, p.ln_g, p.ln_b, w.pre_ln, w.out, w.mean, w.rstd, hp, site_seed(seed, layer, site0 + 1)));
  return IMT_OK;
}

// backward of the block: dy = grad of w.out (in sw.d_run or the caller's d_out); writes the gradient of x into
// sw.d_run; cross: also d_kv_src (accumulate flag).  `slot` selects the dpre/ddrop scratch pair (1 cross, 2 self).
int attn_block_bwd(const Ctx& c, const imt_attn_block& p, AttnWs& w, StackWs& sw, DeferredDW& dw, int slot, const void* x, int B, int T,
                   const void* kv_src, int Tk, const MaskSet& ms, bool training, uint64_t seed, int layer, int site0, const void* dy,
                   void* d_kv_src, int accumulate_kv, int64_t kv_w = -1, int64_t kv_b = -1, void* d_kv_batched = nullptr) {
  const int d = c.m->d;
  const int N = B * T;
  const float hp = training ? c.m->hidden_dropout : 0.f, ap = training ? c.m->attn_dropout : 0.f;
  void* d_pre = sw.dpre[slot];
  void* d_dense = (hp > 0.f) ? sw.ddrop[slot] : d_pre;
  RC(imt_layernorm_bwd(c.dtype, dy, w.pre_ln, c.P(p.ln_g), w.mean, w.rstd, d_pre, c.G(p.ln_g), c.G(p.ln_b), N, d, 0.f, 0,
                       (hp > 0.f) ? d_dense : nullptr, hp, site_seed(seed, layer, site0 + 1), sw.ln_site(3 * layer + slot, d), c.st));
  dw.add(c, d_dense, d, w.ctx, d, N, d, d, p.o_w, p.o_b);
  RC(linear_bwd_input(c, d_dense, d, N, d, p.o_w, d, sw.d_ctx, d, nullptr, 0, nullptr, IMT_AUX_NONE, 0));
  imt_attn_args a;
  if (!kv_src) {
    attn_args(c, a, B, T, T, w.qkv, 3 * d, offp(w.qkv, d, c.es), 3 * d, offp(w.qkv, 2 * d, c.es), 3 * d, w.ctx, w.lse, ms, ap,
              site_seed(seed, layer, site0));
    a.dO = sw.d_ctx; a.lddo = d;
    a.dQ = sw.d_qkv; a.lddq = 3 * d; a.dK = offp(sw.d_qkv, d, c.es); a.lddk = 3 * d; a.dV = offp(sw.d_qkv, 2 * d, c.es); a.lddv = 3 * d;
    a.delta = sw.delta;
    RC(imt_attention_bwd(&a, c.st));
    dw.add(c, sw.d_qkv, 3 * d, x, d, N, 3 * d, d, p.qkv_w, p.qkv_b);
    RC(linear_bwd_input(c, sw.d_qkv, 3 * d, N, 3 * d, p.qkv_w, d, sw.d_run, d, d_pre, d, nullptr, IMT_AUX_NONE, 0));
  } else {
    const int Nk = B * Tk;
    attn_args(c, a, B, T, Tk, w.qkv, d, w.kv, w.kv_ld, offp(w.kv, d, c.es), w.kv_ld, w.ctx, w.lse, ms, ap, site_seed(seed, layer, site0));
    a.dO = sw.d_ctx; a.lddo = d;
    // batched key|value projections: dK|dV of this layer go to its column block of d_kv_all; d(encoder states) and the
    // weight gradients of all layers are formed after the layer loop (imt_stack_backward)
    void* dkv = d_kv_batched ? d_kv_batched : sw.d_kv;
    const int64_t lddkv = d_kv_batched ? w.kv_ld : 2 * d;
    a.dQ = sw.d_q; a.lddq = d; a.dK = dkv; a.lddk = lddkv; a.dV = offp(dkv, d, c.es); a.lddv = lddkv;
    a.delta = sw.delta;
    RC(imt_attention_bwd(&a, c.st));
    dw.add(c, sw.d_q, d, x, d, N, d, d, p.qkv_w, p.qkv_b);
    RC(linear_bwd_input(c, sw.d_q, d, N, d, p.qkv_w, d, sw.d_run, d, d_pre, d, nullptr, IMT_AUX_NONE, 0));
    // the key|value weight gradient stays in this layer's grouped launch either way (its 32 tiles fill CUs the other six
    // products of a decoder layer leave idle; as a separate batched launch it cost +0.07 ms/step)
    dw.add(c, dkv, lddkv, kv_src, d, Nk, 2 * d, d, kv_w, kv_b);
    if (!d_kv_batched) {
      if (d_kv_src)
        RC(linear_bwd_input(c, sw.d_kv, 2 * d, Nk, 2 * d, kv_w, d, d_kv_src, d, nullptr, 0, nullptr, IMT_AUX_NONE, accumulate_kv));
    }
  }
  return IMT_OK;
}

int ffn_fwd(const Ctx& c, const imt_layer_desc& p, LayerWs& w, const void* x, int N, bool training, uint64_t seed, int layer) {
  const int d = c.m->d, ff = c.m->ff;
  const float hp = training ? c.m->hidden_dropout : 0.f;
  RC(linear_fwd(c, x, d, N, d, p.ff1_w, p.ff1_b, ff, w.h, ff, nullptr, 0, w.z, IMT_AUX_GELU_FWD, 0.f, 0));
  RC(dense_resid_ln(c, w.h, ff, N, ff, p.ff2_w, p.ff2_b, d, x, p.ln2_g, p.ln2_b, w.pre_ln2, w.out, w.mean2, w.rstd2, hp, site_seed(seed, layer, 8)));
  return IMT_OK;
}

int ffn_bwd(const Ctx& c, const imt_layer_desc& p, LayerWs& w, StackWs& sw, DeferredDW& dw, const void* x, int N, bool training,
            uint64_t seed, int layer, const void* dy) {
  const int d = c.m->d, ff = c.m->ff;
  const float hp = training ? c.m->hidden_dropout : 0.f;
  void* d_pre = sw.dpre[0];
  void* d_dense = (hp > 0.f) ? sw.ddrop[0] : d_pre;
  RC(imt_layernorm_bwd(c.dtype, dy, w.pre_ln2, c.P(p.ln2_g), w.mean2, w.rstd2, d_pre, c.G(p.ln2_g), c.G(p.ln2_b), N, d, 0.f, 0,
                       (hp > 0.f) ? d_dense : nullptr, hp, site_seed(seed, layer, 8), sw.ln_site(3 * layer, d), c.st));
  dw.add(c, d_dense, d, w.h, ff, N, d, ff, p.ff2_w, p.ff2_b);
  RC(linear_bwd_input(c, d_dense, d, N, d, p.ff2_w, ff, sw.d_ff, ff, nullptr, 0, w.z, IMT_AUX_DGELU, 0));  // dz
  dw.add(c, sw.d_ff, ff, x, d, N, ff, d, p.ff1_w, p.ff1_b);
  RC(linear_bwd_input(c, sw.d_ff, ff, N, ff, p.ff1_w, d, sw.d_run, d, d_pre, d, nullptr, IMT_AUX_NONE, 0));
  return IMT_OK;
}

int validate(const imt_stack_desc* m, const imt_stack_io* io, const void* ws, int64_t ws_bytes, StackWs& w, LayerWs* layers) {
  IMT_CHECK_ARG(m && io, "stack: null descriptor");
  IMT_CHECK_ARG(m->dtype == IMT_F32 || m->dtype == IMT_BF16, "stack: bad dtype");
  IMT_CHECK_ARG(m->n_layers >= 0 && m->n_layers <= MAX_LAYERS && (m->n_layers == 0 || m->layers), "stack: bad layer table");
  IMT_CHECK_ARG(m->d > 0 && m->heads > 0 && m->d % m->heads == 0, "stack: hidden size %d not a multiple of heads %d", m->d, m->heads);
  const int dh = m->d / m->heads;
  IMT_CHECK_ARG(dh == 32 || dh == 64, "stack: head_dim %d unsupported (32 or 64)", dh);
  IMT_CHECK_ARG(m->d % 8 == 0 && m->ff % 8 == 0, "stack: d and ff must be multiples of 8");
  IMT_CHECK_ARG(io->B > 0 && io->T > 0, "stack: empty batch");
  IMT_CHECK_ARG(io->T <= m->max_pos || io->pos_ids, "stack: sequence longer than max_position_embeddings");
  IMT_CHECK_ARG(m->params && io->ids && io->out, "stack: null tensor");
  if (m->is_decoder) IMT_CHECK_ARG(io->enc_states && io->Tk > 0, "stack: decoder needs encoder states");
  carve(m, io->B, io->T, m->is_decoder ? io->Tk : 0, const_cast<void*>(ws), w, layers);
  IMT_CHECK_ARG(ws && ws_bytes >= w.bytes, "stack: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)w.bytes);
  IMT_CHECK_ARG(((uintptr_t)ws & 255) == 0, "stack: workspace must be 256-B aligned");
  return IMT_OK;
}

}  // namespace

extern "C" int64_t imt_stack_workspace_bytes(const imt_stack_desc* m, int B, int T, int Tk) {
  if (!m || m->n_layers > MAX_LAYERS || (m->n_layers > 0 && !m->layers)) return -1;
  StackWs w; LayerWs layers[MAX_LAYERS];
  carve(m, B, T, m->is_decoder ? Tk : 0, nullptr, w, layers);
  return w.bytes;
}

extern "C" int imt_stack_forward(const imt_stack_desc* m, const imt_stack_io* io, void* ws, int64_t ws_bytes, void* stream) {
  StackWs w; LayerWs layers[MAX_LAYERS];
  RC(validate(m, io, ws, ws_bytes, w, layers));
  Ctx c{m, (hipStream_t)stream, m->dtype, esize(m->dtype)};
  c.splitk = w.splitk; c.splitk_bytes = w.splitk_bytes;
  const int B = io->B, T = io->T, N = B * T, d = m->d;
  const bool training = io->training != 0;
  const uint64_t seed = io->dropout_seed;
  void* x0 = (m->n_layers == 0) ? io->out : w.x0;
  // row-block tickets of the in-launch LayerNorm: zero on entry of every launch that uses them (each leaves them zero;
  // the workspace itself arrives uninitialised)
  if (m->n_layers > 0 && imt_gemm_ln_ticket_enabled()) {
    if (hipMemsetAsync(w.ln_tickets, 0, w.ln_ticket_bytes, c.st) != hipSuccess) { imt_set_error("stack forward: memset of the LayerNorm tickets failed"); return IMT_ERR_LAUNCH; }
    c.ln_tickets = w.ln_tickets;
  }
  // parameters still being written by an optimizer step on another stream: wait site by site (imt_stack_io.wait_events)
  auto wait_site = [&](int k) {
    if (io->wait_events && k < io->n_wait_events && io->wait_events[k])
      (void)hipStreamWaitEvent(c.st, (hipEvent_t)const_cast<void*>(io->wait_events[k]), 0);
  };
  wait_site(0);
  // BertEmbeddings: gather + sum + LayerNorm + dropout in one launch (the sum is kept for the backward)
  RC(imt_embed_ln_fwd(c.dtype, io->ids, io->pos_ids, io->type_ids, c.P(m->emb_word), c.P(m->emb_pos), c.P(m->emb_type), c.P(m->emb_ln_g),
                      c.P(m->emb_ln_b), w.emb_sum, x0, w.emb_mean, w.emb_rstd, N, T, d, m->vocab, m->max_pos, m->n_types, m->ln_eps,
                      training ? m->hidden_dropout : 0.f, site_seed(seed, 1000, 0), c.st));
  const void* x = x0;
  const MaskSet self_ms{io->key_mask, io->query_mask, io->mask3d, io->causal};
  const MaskSet cross_ms{io->enc_mask, nullptr, nullptr, 0};
  const bool batched = cross_kv_batched(m);
  if (batched)  // key|value projections of the encoder states for ALL layers: one [B*Tk, d] x [L*2d, d]^T product
    RC(linear_fwd(c, io->enc_states, d, B * io->Tk, d, m->layers[0].cross_kv_w, m->layers[0].cross_kv_b, m->n_layers * 2 * d, w.kv_all,
                  (int64_t)m->n_layers * 2 * d, nullptr, 0, nullptr, IMT_AUX_NONE, 0.f, 0));
  for (int l = 0; l < m->n_layers; ++l) {
    const imt_layer_desc& p = m->layers[l];
    LayerWs& L = layers[l];
    wait_site(1 + l);
    RC(attn_block_fwd(c, p.self_attn, L.self_attn, x, B, T, nullptr, 0, self_ms, training, seed, l, 0));
    const void* a = L.self_attn.out;
    if (m->is_decoder && p.cross_attn.qkv_w >= 0) {
      RC(attn_block_fwd(c, p.cross_attn, L.cross, a, B, T, io->enc_states, io->Tk, cross_ms, training, seed, l, 4, kv_w_off(m, p),
                        kv_b_off(m, p), batched));
      a = L.cross.out;
    }
    if (l == m->n_layers - 1) L.out = io->out;  // last LN writes straight into the caller's output
    RC(ffn_fwd(c, p, L, a, N, training, seed, l));
    x = L.out;
  }
  return IMT_OK;
}

extern "C" int imt_stack_backward(const imt_stack_desc* m, const imt_stack_io* io, void* ws, int64_t ws_bytes, int layer_lo,
                                  int layer_hi, void* stream) {
  StackWs w; LayerWs layers[MAX_LAYERS];
  RC(validate(m, io, ws, ws_bytes, w, layers));
  IMT_CHECK_ARG(m->grads && io->d_out, "stack_backward: grads / d_out missing");
  IMT_CHECK_ARG(0 <= layer_lo && layer_lo <= layer_hi && layer_hi <= m->n_layers, "stack_backward: bad layer range");
  Ctx c{m, (hipStream_t)stream, m->dtype, esize(m->dtype)};
  c.splitk = w.splitk; c.splitk_bytes = w.splitk_bytes;
  const int B = io->B, T = io->T, N = B * T, d = m->d;
  const bool training = io->training != 0;
  const uint64_t seed = io->dropout_seed;
  const MaskSet self_ms{io->key_mask, io->query_mask, io->mask3d, io->causal};
  const MaskSet cross_ms{io->enc_mask, nullptr, nullptr, 0};
  // The running gradient w.r.t. the current layer's output lives in w.d_run between layers (and between segment
  // calls); the first segment reads it from io->d_out.
  const void* dy = (layer_hi == m->n_layers) ? io->d_out : w.d_run;
  const bool batched = cross_kv_batched(m);
  DeferredDW dw;
  // LayerNorm dgamma / dbeta of this call's layers go to per-XCD partial sums (zeroed here, folded into the gradients by
  // one launch at the end): 256 workgroups per LayerNorm adding straight into the same d gradient addresses cost ~6 us
  // of serialised atomics per launch
  const int64_t lnf = IMT_LN_BWD_WS_FLOATS(d);
  if (layer_hi > layer_lo) (void)hipMemsetAsync(w.ln_site(3 * layer_lo, d), 0, (size_t)(3 * (layer_hi - layer_lo)) * lnf * 4, c.st);
  if (layer_lo == 0) (void)hipMemsetAsync(w.ln_site(3 * m->n_layers, d), 0, (size_t)lnf * 4, c.st);
  // weight gradients on the side stream: only when this call covers the whole stack (a data-parallel run calls per layer
  // and hands each layer's gradients to its all-reduce bucket as soon as the call returns) and IMT_DW_SIDE_STREAM != 0
  static const bool side_env = !(getenv("IMT_DW_SIDE_STREAM") && atoi(getenv("IMT_DW_SIDE_STREAM")) == 0);
  // (not while the per-launch event profiler is on: a launch timed next to a concurrent one is not that kernel's time)
  const bool side = side_env && !imt_prof_enabled() && layer_lo == 0 && layer_hi == m->n_layers && m->n_layers >= 2 && g_side.init();
  int last_side = -1;
  for (int l = layer_hi - 1; l >= layer_lo; --l) {
    const imt_layer_desc& p = m->layers[l];
    LayerWs& L = layers[l];
    w.use_scratch(l);
    // this layer writes the scratch set dW(l+2) read from: that launch must be over (it is, by a layer's worth of time)
    if (side && l + 2 < layer_hi) (void)hipStreamWaitEvent(c.st, g_side.done[l + 2], 0);
    if (l == m->n_layers - 1) L.out = io->out;
    const bool has_cross = m->is_decoder && p.cross_attn.qkv_w >= 0;
    const void* x_in = (l == 0) ? w.x0 : (const void*)layers[l - 1].out;
    const void* a_self = L.self_attn.out;
    const void* a_ffn_in = has_cross ? L.cross.out : a_self;
    RC(ffn_bwd(c, p, L, w, dw, a_ffn_in, N, training, seed, l, dy));
    if (has_cross) {
      const bool first_cross = (l == m->n_layers - 1);
      void* dkv_l = batched ? offp(w.d_kv_all, (int64_t)l * 2 * d, c.es) : nullptr;
      RC(attn_block_bwd(c, p.cross_attn, L.cross, w, dw, 1, a_self, B, T, io->enc_states, io->Tk, cross_ms, training, seed, l, 4, w.d_run,
                        io->d_enc_states, first_cross ? 0 : 1, kv_w_off(m, p), kv_b_off(m, p), dkv_l));
    }
    RC(attn_block_bwd(c, p.self_attn, L.self_attn, w, dw, 2, x_in, B, T, nullptr, 0, self_ms, training, seed, l, 0, w.d_run, nullptr, 0));
    // all 4 (encoder) / 7 (decoder) weight-gradient GEMMs of this layer: one launch
    if (side) {
      (void)hipEventRecord(g_side.ready[l], c.st);
      (void)hipStreamWaitEvent(g_side.st, g_side.ready[l], 0);
      RC(dw.flush(g_side.st));
      (void)hipEventRecord(g_side.done[l], g_side.st);
      last_side = l;
    } else {
      RC(dw.flush(c.st));
    }
    dy = w.d_run;
  }
  // the gradients are complete, in main-stream order, when this call returns (the encoder stack accumulates into the
  // self-attention weights it shares with the decoder; the optimizer / all-reduce follow on the main stream)
  // (joined at the end of this function)
  if (layer_lo == 0 && batched) {
    // every layer has written its dK|dV block: d(encoder states) = d_kv_all [Nk, L*2d] x W_kv [L*2d, d], one product with
    // K = L*2d instead of L accumulating ones
    const int Nk = B * io->Tk;
    const int L2d = m->n_layers * 2 * d;
    if (io->d_enc_states)
      RC(linear_bwd_input(c, w.d_kv_all, L2d, Nk, L2d, m->layers[0].cross_kv_w, d, io->d_enc_states, d, nullptr, 0, nullptr, IMT_AUX_NONE, 0));
  }
  if (layer_lo == 0) {
    const void* dy0 = (m->n_layers == 0) ? io->d_out : dy;
    // (w.d_emb, not a scratch buffer of the layers: dW of layers 0 / 1 may still be reading those on the side stream)
    RC(imt_layernorm_bwd(c.dtype, dy0, w.emb_sum, c.P(m->emb_ln_g), w.emb_mean, w.emb_rstd, w.d_emb, c.G(m->emb_ln_g), c.G(m->emb_ln_b), N, d,
                         training ? m->hidden_dropout : 0.f, site_seed(seed, 1000, 0), nullptr, 0.f, 0, w.ln_site(3 * m->n_layers, d), c.st));
    RC(imt_embed_bwd(c.dtype, io->ids, io->pos_ids, io->type_ids, w.d_emb, c.G(m->emb_word), c.G(m->emb_pos), c.G(m->emb_type), N, T, d,
                     m->pad_id, c.st));
  }
  {
    // fold the partial sums of the sites this call covered: [3*layer_lo, 3*layer_hi) and, with layer 0, the embeddings
    static int64_t g_off[3 * MAX_LAYERS + 1], b_off[3 * MAX_LAYERS + 1];
    const int first = 3 * layer_lo, emb = 3 * m->n_layers;
    const int last = (layer_lo == 0) ? emb + 1 : 3 * layer_hi;  // [first, last) -- sites of layers above layer_hi are skipped
    for (int sidx = first; sidx < last; ++sidx) {
      g_off[sidx - first] = -1; b_off[sidx - first] = -1;
      if (sidx == emb) { g_off[sidx - first] = m->emb_ln_g; b_off[sidx - first] = m->emb_ln_b; continue; }
      const int l = sidx / 3, slot = sidx % 3;
      if (l >= layer_hi) continue;
      const imt_layer_desc& p = m->layers[l];
      if (slot == 0) { g_off[sidx - first] = p.ln2_g; b_off[sidx - first] = p.ln2_b; }
      else if (slot == 2) { g_off[sidx - first] = p.self_attn.ln_g; b_off[sidx - first] = p.self_attn.ln_b; }
      else if (m->is_decoder && p.cross_attn.qkv_w >= 0) { g_off[sidx - first] = p.cross_attn.ln_g; b_off[sidx - first] = p.cross_attn.ln_b; }
    }
    if (last > first) RC(imt_ln_partial_reduce(w.ln_site(first, d), last - first, d, g_off, b_off, m->grads, c.st));
  }
  if (last_side >= 0) (void)hipStreamWaitEvent(c.st, g_side.done[last_side], 0);
  return IMT_OK;
}

// ------------------------------------------------------------------------------------------------ incremental decoding
namespace {

struct DecodeWs {
  void* emb_sum; void* x; void* ctx; void* pre_ln; void* a; void* q; void* b; void* h; void* z; float* mean; float* rstd;
  void* splitk; int64_t splitk_bytes;  // imt_gemm's split-K slabs (R rows: a handful of output tiles per product)
  char* fused; int64_t fused_layer_bytes; unsigned* fused_bar;  // one-launch step (decode_fused.hip): per-layer hand-off buffers, barrier words
  int64_t bytes;
};

// the one-launch decoder step (decode_fused.hip): bf16, hidden size 512 or 768 in heads of 64
bool fused_decode_shape(const imt_stack_desc* m) {
  return m->dtype == IMT_BF16 && imt_decode_fused_shape(m->d, m->heads, m->ff, m->n_layers);
}

void carve_decode(const imt_stack_desc* m, int r_max, void* ws, DecodeWs& w) {
  Carver c(ws);
  const int64_t R = r_max, d = m->d, ff = m->ff, es = esize(m->dtype);
  w.emb_sum = c.take(R * d * es); w.x = c.take(R * d * es); w.ctx = c.take(R * d * es); w.pre_ln = c.take(R * d * es);
  w.a = c.take(R * d * es); w.q = c.take(R * d * es); w.b = c.take(R * d * es);
  w.h = c.take(R * ff * es); w.z = c.take(R * ff * es);
  w.mean = (float*)c.take(R * 4); w.rstd = (float*)c.take(R * 4);
  w.splitk_bytes = imt_gemm_splitk_ws_bytes();
  w.splitk = c.take(w.splitk_bytes);
  w.fused = nullptr; w.fused_layer_bytes = 0; w.fused_bar = nullptr;
  if (fused_decode_shape(m)) {
    w.fused_layer_bytes = (imt_decode_fused_layer_bytes(r_max, m->d, m->ff) + 255) & ~(int64_t)255;
    w.fused = (char*)c.take(w.fused_layer_bytes * m->n_layers);
    w.fused_bar = (unsigned*)c.take(IMT_FUSED_BAR_WORDS * sizeof(unsigned));
  }
  w.bytes = c.off;
}

int validate_decoder(const imt_stack_desc* m) {
  IMT_CHECK_ARG(m, "decode: null descriptor");
  IMT_CHECK_ARG(m->dtype == IMT_F32 || m->dtype == IMT_BF16, "decode: bad dtype");
  IMT_CHECK_ARG(m->is_decoder, "decode: stack is not a decoder");
  IMT_CHECK_ARG(m->n_layers > 0 && m->n_layers <= MAX_LAYERS && m->layers, "decode: bad layer table");
  IMT_CHECK_ARG(m->d > 0 && m->heads > 0 && m->d % m->heads == 0, "decode: hidden size %d not a multiple of heads %d", m->d, m->heads);
  const int dh = m->d / m->heads;
  IMT_CHECK_ARG(dh == 32 || dh == 64, "decode: head_dim %d unsupported (32 or 64)", dh);
  IMT_CHECK_ARG(m->d % 8 == 0 && m->ff % 8 == 0, "decode: d and ff must be multiples of 8");
  IMT_CHECK_ARG(m->params, "decode: null parameter buffer");
  for (int l = 0; l < m->n_layers; ++l)
    IMT_CHECK_ARG(m->layers[l].cross_attn.qkv_w >= 0, "decode: layer %d has no crossattention block", l);
  return IMT_OK;
}

}  // namespace

extern "C" int64_t imt_decode_workspace_bytes(const imt_stack_desc* m, int r_max) {
  if (!m || r_max <= 0) return -1;
  DecodeWs w;
  carve_decode(m, r_max, nullptr, w);
  return w.bytes;
}
extern "C" int64_t imt_decode_self_cache_bytes(const imt_stack_desc* m, int r_max, int t_max) {
  if (!m || r_max <= 0 || t_max <= 0) return -1;
  return (int64_t)m->n_layers * r_max * t_max * 3 * m->d * esize(m->dtype);
}
extern "C" int64_t imt_decode_cross_bytes(const imt_stack_desc* m, int B, int Tk) {
  if (!m || B <= 0 || Tk <= 0) return -1;
  return (int64_t)m->n_layers * B * Tk * 2 * m->d * esize(m->dtype);
}

extern "C" int imt_decode_begin(const imt_stack_desc* m, const void* enc_states, int B, int Tk, void* cross_kv, void* stream) {
  RC(validate_decoder(m));
  IMT_CHECK_ARG(enc_states && cross_kv && B > 0 && Tk > 0, "decode_begin: bad arguments");
  Ctx c{m, (hipStream_t)stream, m->dtype, esize(m->dtype)};
  const int d = m->d;
  const int64_t per_layer = (int64_t)B * Tk * 2 * d;
  for (int l = 0; l < m->n_layers; ++l) {
    const imt_attn_block& p = m->layers[l].cross_attn;
    RC(linear_fwd(c, enc_states, d, B * Tk, d, kv_w_off(m, m->layers[l]), kv_b_off(m, m->layers[l]), 2 * d,
                  offp(cross_kv, l * per_layer, c.es), 2 * d, nullptr, 0, nullptr, IMT_AUX_NONE, 0.f, 0));
  }
  return IMT_OK;
}

extern "C" int imt_decode_check(const imt_stack_desc* m, int r_max, const void* ws, void* stream) {
  RC(validate_decoder(m));
  IMT_CHECK_ARG(ws && r_max > 0, "decode_check: bad arguments");
  DecodeWs w;
  carve_decode(m, r_max, const_cast<void*>(ws), w);
  if (!w.fused || !imt_decode_fused_enabled()) return IMT_OK;
  unsigned status = 0;
  if (hipMemcpyAsync(&status, w.fused_bar + IMT_FUSED_BAR_WORDS - 1, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    imt_set_error("decode_check: cannot read the status word");
    return IMT_ERR_LAUNCH;
  }
  if (status != 0) {
    imt_set_error("decode: a one-launch decoder step was abandoned (grid barrier %u timed out); its tokens are not valid", status & 0xfffu);
    return IMT_ERR_LAUNCH;
  }
  return IMT_OK;
}

extern "C" int imt_decode_step(const imt_stack_desc* m, const imt_decode_io* io, void* ws, int64_t ws_bytes, void* stream) {
  RC(validate_decoder(m));
  IMT_CHECK_ARG(io, "decode_step: null io");
  IMT_CHECK_ARG(io->R > 0 && io->rep > 0 && io->R % io->rep == 0 && io->R <= io->r_max, "decode_step: bad row counts (R=%d rep=%d r_max=%d)", io->R, io->rep, io->r_max);
  IMT_CHECK_ARG(io->pos >= 0 && io->pos < io->t_max && io->Tk > 0, "decode_step: position %d outside the cache (t_max=%d)", io->pos, io->t_max);
  IMT_CHECK_ARG(io->ids && io->pos_ids && io->self_cache && io->cross_kv && io->out, "decode_step: null tensor");
  DecodeWs w;
  carve_decode(m, io->r_max, ws, w);
  IMT_CHECK_ARG(ws && ws_bytes >= w.bytes, "decode_step: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)w.bytes);
  IMT_CHECK_ARG(((uintptr_t)ws & 255) == 0, "decode_step: workspace must be 256-B aligned");
  Ctx c{m, (hipStream_t)stream, m->dtype, esize(m->dtype)};
  c.splitk = w.splitk; c.splitk_bytes = w.splitk_bytes;
  const int R = io->R, d = m->d, ff = m->ff, H = m->heads, dh = d / H;
  const int64_t row3 = (int64_t)io->t_max * 3 * d;                       // one cache row (all positions)
  const int64_t self_layer = (int64_t)io->r_max * row3;
  const int B = R / io->rep;
  const int64_t cross_layer = (int64_t)B * io->Tk * 2 * d;
  if (w.fused && imt_decode_fused_enabled()) {
    // one launch for the whole stack (decode_fused.hip); the per-operator chain below stays for fp32 and other shapes
    ImtFusedArgs f;
    memset(&f, 0, sizeof(f));
    const bf16_t* P = reinterpret_cast<const bf16_t*>(m->params);
    for (int l = 0; l < m->n_layers; ++l) {
      const imt_layer_desc& p = m->layers[l];
      ImtFusedLayer& L = f.L[l];
      L.wqkv = P + p.self_attn.qkv_w; L.bqkv = P + p.self_attn.qkv_b; L.wo = P + p.self_attn.o_w; L.bo = P + p.self_attn.o_b;
      L.g1 = P + p.self_attn.ln_g; L.b1 = P + p.self_attn.ln_b;
      L.wq = P + p.cross_attn.qkv_w; L.bq = P + p.cross_attn.qkv_b; L.wo2 = P + p.cross_attn.o_w; L.bo2 = P + p.cross_attn.o_b;
      L.g2 = P + p.cross_attn.ln_g; L.b2 = P + p.cross_attn.ln_b;
      L.w1 = P + p.ff1_w; L.bf1 = P + p.ff1_b; L.w2 = P + p.ff2_w; L.bf2 = P + p.ff2_b; L.g3 = P + p.ln2_g; L.b3 = P + p.ln2_b;
      L.cache = reinterpret_cast<bf16_t*>(offp(io->self_cache, l * self_layer, c.es));
      L.cross_kv = reinterpret_cast<const bf16_t*>(offp(io->cross_kv, l * cross_layer, c.es));
      Carver fc(w.fused + l * w.fused_layer_bytes);
      const int64_t rm = io->r_max;
      L.xin = (bf16_t*)fc.take(rm * d * 2); L.ctx1 = (bf16_t*)fc.take(rm * d * 2); L.a = (bf16_t*)fc.take(rm * d * 2);
      L.q = (bf16_t*)fc.take(rm * d * 2); L.ctx2 = (bf16_t*)fc.take(rm * d * 2); L.b = (bf16_t*)fc.take(rm * d * 2);
      L.h = (bf16_t*)fc.take(rm * ff * 2);
      L.pre1 = (float*)fc.take(rm * d * 4); L.pre2 = (float*)fc.take(rm * d * 4); L.pre3 = (float*)fc.take(rm * d * 4);
    }
    f.n_layers = m->n_layers; f.d = d; f.R = R; f.rep = io->rep; f.pos = io->pos; f.Tk = io->Tk; f.t_max = io->t_max; f.r_max = io->r_max;
    f.H = H; f.dh = dh; f.ff = ff;
    f.ids = io->ids; f.pos_ids = io->pos_ids; f.type_ids = io->type_ids;
    f.emb_word = P + m->emb_word; f.emb_pos = P + m->emb_pos; f.emb_type = P + m->emb_type; f.emb_g = P + m->emb_ln_g; f.emb_b = P + m->emb_ln_b;
    f.vocab = m->vocab; f.max_pos = m->max_pos; f.n_types = m->n_types;
    f.out = reinterpret_cast<bf16_t*>(io->out);
    f.slots = io->slots; f.enc_mask = io->enc_mask; f.eps = m->ln_eps; f.bar = w.fused_bar;
    if (io->pos == 0 && hipMemsetAsync(w.fused_bar + IMT_FUSED_BAR_WORDS - 1, 0, sizeof(unsigned), c.st) != hipSuccess) {
      imt_set_error("decode_step: memset failed");   // a new search: clear the sticky status word
      return IMT_ERR_LAUNCH;
    }
    return imt_decode_fused_launch(f, c.st);
  }
  RC(imt_embed_ln_fwd(c.dtype, io->ids, io->pos_ids, io->type_ids, c.P(m->emb_word), c.P(m->emb_pos), c.P(m->emb_type), c.P(m->emb_ln_g),
                      c.P(m->emb_ln_b), w.emb_sum, w.x, w.mean, w.rstd, R, 1, d, m->vocab, m->max_pos, m->n_types, m->ln_eps, 0.f, 0, c.st));
  const void* x = w.x;
  for (int l = 0; l < m->n_layers; ++l) {
    const imt_layer_desc& p = m->layers[l];
    // self attention: q|k|v of the new position straight into the cache row of each hypothesis
    void* cache_l = offp(io->self_cache, l * self_layer, c.es);
    void* qkv_new = offp(cache_l, (int64_t)io->pos * 3 * d, c.es);
    RC(linear_fwd(c, x, d, R, d, p.self_attn.qkv_w, p.self_attn.qkv_b, 3 * d, qkv_new, row3, nullptr, 0, nullptr, IMT_AUX_NONE, 0.f, 0));
    imt_attn_decode_args a;
    memset(&a, 0, sizeof(a));
    a.dtype = c.dtype; a.R = R; a.H = H; a.head_dim = dh; a.n_keys = io->pos + 1; a.rep = 1;
    a.Q = qkv_new; a.ldq = row3;
    a.K = offp(cache_l, d, c.es); a.V = offp(cache_l, 2 * d, c.es); a.ld_row = row3; a.ld_pos = 3 * d;
    a.slots = io->slots; a.ld_slots = io->t_max;
    a.O = w.ctx; a.ldo = d; a.scale = 1.0f / sqrtf((float)dh);
    RC(imt_attention_decode(&a, c.st));
    RC(dense_resid_ln(c, w.ctx, d, R, d, p.self_attn.o_w, p.self_attn.o_b, d, x, p.self_attn.ln_g, p.self_attn.ln_b, w.pre_ln, w.a, w.mean, w.rstd, 0.f, 0));
    // cross attention against the per-sentence K|V
    RC(linear_fwd(c, w.a, d, R, d, p.cross_attn.qkv_w, p.cross_attn.qkv_b, d, w.q, d, nullptr, 0, nullptr, IMT_AUX_NONE, 0.f, 0));
    const void* kv_l = offp(io->cross_kv, l * cross_layer, c.es);
    memset(&a, 0, sizeof(a));
    a.dtype = c.dtype; a.R = R; a.H = H; a.head_dim = dh; a.n_keys = io->Tk; a.rep = io->rep;
    a.Q = w.q; a.ldq = d;
    a.K = kv_l; a.V = offp(kv_l, d, c.es); a.ld_row = (int64_t)io->Tk * 2 * d; a.ld_pos = 2 * d;
    a.key_mask = io->enc_mask; a.ld_mask = io->Tk;
    a.O = w.ctx; a.ldo = d; a.scale = 1.0f / sqrtf((float)dh);
    RC(imt_attention_decode(&a, c.st));
    RC(dense_resid_ln(c, w.ctx, d, R, d, p.cross_attn.o_w, p.cross_attn.o_b, d, w.a, p.cross_attn.ln_g, p.cross_attn.ln_b, w.pre_ln, w.b, w.mean, w.rstd, 0.f, 0));
    // feed-forward
    RC(linear_fwd(c, w.b, d, R, d, p.ff1_w, p.ff1_b, ff, w.h, ff, nullptr, 0, w.z, IMT_AUX_GELU_FWD, 0.f, 0));
    void* y = (l == m->n_layers - 1) ? io->out : w.x;
    RC(dense_resid_ln(c, w.h, ff, R, ff, p.ff2_w, p.ff2_b, d, w.b, p.ln2_g, p.ln2_b, w.pre_ln, y, w.mean, w.rstd, 0.f, 0));
    x = y;
  }
  return IMT_OK;
}
