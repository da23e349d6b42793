// HBM-bound row kernels of the path: LayerNorm fwd/bwd, embedding gather / scatter-add, column sums
// (bias gradients), row gather/scatter (non-pad target row select), casts.
// One 64-lane wave owns one row; every lane moves 4 consecutive elements per access (8 B bf16 / 16 B fp32),
// so a wave instruction covers 256 consecutive elements; row statistics are wave reductions.
#include <stdlib.h>
#include "common.hpp"

namespace {

constexpr int ROWS_PER_BLOCK = 4;  // 4 waves

// ------------------------------------------------------------------------------------------- LayerNorm fwd
// Where the rows come from: SRC 0 = x ; 1 = x + resid (the sum is also written: LayerNorm's backward reads it) ;
// 2 = word[ids] + pos[pos_ids | n % seq_len] + type[type_ids] (HF BertEmbeddings: gather + LayerNorm + dropout in one launch,
// the sum written for the backward).
template <typename T> struct LnSrc {
  const T* x; const T* resid; T* sum_out;
  const int64_t* ids; const int64_t* pos_ids; const int64_t* type_ids;
  const T* word; const T* pos; const T* type;
  int seq_len, vocab, max_pos, n_types;
};

template <typename T, int NCH, int SRC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnSrc<T> src, const T* __restrict__ gamma,
                                                     const T* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int rows, int d, float eps, uint32_t thresh, float inv_keep,
                                                     uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int row = imt_xcd_block(blockIdx.x, gridDim.x) * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  f32x4 v[NCH], gv[NCH], bv[NCH];
  float s = 0.f;
  // every load of the row is requested up front and unconditionally (columns past d read a clamped address and are
  // masked where used): a load under a branch makes the compiler drain vmcnt at the join, one round trip per load
  if (SRC == 2) {
    int64_t wi = src.ids[row];
    int64_t pi = src.pos_ids ? src.pos_ids[row] : (int64_t)(row % src.seq_len);
    int64_t ti = src.type_ids ? src.type_ids[row] : 0;
    // clamp like a defensive gather: out-of-range ids would fault in the reference; here they read row 0
    if (wi < 0 || wi >= src.vocab) wi = 0;
    if (pi < 0 || pi >= src.max_pos) pi = 0;
    if (ti < 0 || ti >= src.n_types) ti = 0;
    const T* wr = src.word + wi * d;
    const T* pr = src.pos + pi * d;
    const T* tr = src.type + ti * d;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = min(lane * 4 + i * 256, d - 4);
      const f32x4 a = Vec4<T>::load(wr + c), b = Vec4<T>::load(pr + c), t = Vec4<T>::load(tr + c);
      gv[i] = Vec4<T>::load(gamma + c);
      bv[i] = Vec4<T>::load(beta + c);
      v[i] = a + b;
      v[i] += t;
    }
  } else {
    const T* xr = src.x + (int64_t)row * d;
    const T* rr = (SRC == 1) ? src.resid + (int64_t)row * d : xr;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = min(lane * 4 + i * 256, d - 4);
      v[i] = Vec4<T>::load(xr + c);
      if (SRC == 1) v[i] += Vec4<T>::load(rr + c);
      gv[i] = Vec4<T>::load(gamma + c);
      bv[i] = Vec4<T>::load(beta + c);
    }
  }
  if (SRC != 0) {  // the stored sum is what the backward (and this LayerNorm) sees: rounded to T
    T* so = src.sum_out + (int64_t)row * d;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane * 4 + i * 256;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = to_f32<T>(from_f32<T>(v[i][e]));
      if (c < d) Vec4<T>::store(so + c, v[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (lane * 4 + i * 256 < d) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < d) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float t = v[i][e] - mean; q += t * t; }
    }
  }
  const float var = wave_sum(q) / (float)d;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  T* yr = y + (int64_t)row * d;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < d) {
      const f32x4 g = gv[i], b = bv[i];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      if (thresh) dropout_apply4(o, seed, (uint64_t)row * d + c, thresh, inv_keep);
      Vec4<T>::store(yr + c, o);
    }
  }
}

// ------------------------------------------------------------------------------------------- LayerNorm bwd
// Each wave walks `rows_per_wave` rows, keeps per-column partial dgamma/dbeta in registers, the WPB waves of
// a block combine through LDS and issue one fp32 atomic per column per block.
template <typename T, int NCH, int WPB>
__global__ __launch_bounds__(WPB * 64, ((WPB == 8 && NCH <= 2 && sizeof(T) == 2) ? 4 : 1)) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const T* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows,
                                                     int d, int rows_per_wave, uint32_t y_thresh, float y_inv_keep,
                                                     uint64_t y_seed, T* __restrict__ dx_drop, uint32_t dx_thresh,
                                                     float dx_inv_keep, uint64_t dx_seed, float* __restrict__ partial,
                                                     unsigned long long* trace) {
  __shared__ float red[2][WPB][NCH * 256];
  IMT_STAMP(trace, 0);  // tuning only (IMT_TRACE=ln_bwd)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  f32x4 g[NCH], ag[NCH], ab[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 4 + i * 256;
    g[i] = Vec4<T>::load(gamma + min(c, d - 4));  // unconditional (clamped): only used under c < d
    ag[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int row_begin = (imt_xcd_block(blockIdx.x, gridDim.x) * WPB + w) * rows_per_wave;
  // rows are processed in batches of RB: all loads of a batch are issued before the first reduction, so a wave keeps
  // 2*RB row loads in flight (with ~1 wave per SIMD the row loop is otherwise a chain of exposed memory latencies)
  constexpr int RB = 4;
  for (int r0 = 0; r0 < rows_per_wave; r0 += RB) {
    f32x4 xb[RB][NCH], db[RB][NCH];
    float mub[RB], rsb[RB];
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      // NO control flow around these loads (a branch makes the compiler drain vmcnt at its join, which serialises the
      // batch into one memory round trip per row -- IMT_LN_TRACE: 7.6 us to issue 4 rows): dead rows / columns read a
      // clamped address instead and are masked where they are used
      const int row = min(row_begin + r0 + k, rows - 1);
      mub[k] = mean[row];
      rsb[k] = rstd[row];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = min(lane * 4 + i * 256, d - 4);
        xb[k][i] = Vec4<T>::load(x + (int64_t)row * d + c);
        db[k][i] = Vec4<T>::load(dy + (int64_t)row * d + c);
      }
    }
    if (r0 == 0) IMT_STAMP(trace, 1);
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      const int row = row_begin + r0 + k;
      if (r0 == 0 && k == 1) IMT_STAMP(trace, 3);
      if (!((r0 + k < rows_per_wave) && (row < rows))) continue;  // wave-uniform
      const float mu = mub[k], rs = rsb[k];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < d) {
          if (y_thresh) dropout_apply4(db[k][i], y_seed, (uint64_t)row * d + c, y_thresh, y_inv_keep);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xb[k][i][e] = (xb[k][i][e] - mu) * rs;
            const float dg = db[k][i][e] * g[i][e];
            s1 += dg;
            s2 += dg * xb[k][i][e];
            ag[i][e] += db[k][i][e] * xb[k][i][e];
            ab[i][e] += db[k][i][e];
          }
        }
      }
      const float c1 = wave_sum(s1) / (float)d, c2 = wave_sum(s2) / (float)d;
      if (r0 == 0 && k == 0) IMT_STAMP(trace, 2);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane * 4 + i * 256;
        if (c < d) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rs * (db[k][i][e] * g[i][e] - c1 - xb[k][i][e] * c2);
          Vec4<T>::store(dx + (int64_t)row * d + c, o);
          if (dx_drop) {
            f32x4 o2 = o;
            if (dx_thresh) dropout_apply4(o2, dx_seed, (uint64_t)row * d + c, dx_thresh, dx_inv_keep);
            Vec4<T>::store(dx_drop + (int64_t)row * d + c, o2);
          }
        }
      }
    }
  }
  IMT_STAMP(trace, 5);
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][w][i * 256 + lane * 4 + e] = ag[i][e];
      red[1][w][i * 256 + lane * 4 + e] = ab[i][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += WPB * 64) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int k = 0; k < WPB; ++k) { sg += red[0][k][c]; sb += red[1][k][c]; }
    // 256 workgroups adding into the same d addresses serialise at the memory side: ~6 us after the last wave has issued
    // its atomics (IMT_LN_NO_ATOMICS experiment: 21 -> 15 us per launch).  With `partial` ([copies][2][d], zeroed by the caller)
    // workgroup b adds into copy b % copies -- one of its own XCD's (b % 8) -- and imt_ln_partial_reduce folds the copies into the gradients.
    if (partial) {
      float* pc = partial + (size_t)(blockIdx.x & (IMT_LN_PARTIAL_COPIES - 1)) * 2 * d;
      atomicAdd(pc + c, sg);
      atomicAdd(pc + d + c, sb);
    } else if (dgamma) {
      atomicAdd(dgamma + c, sg);
      atomicAdd(dbeta + c, sb);
    }
  }
  IMT_STAMP(trace, 6);
}

// ------------------------------------------------------------------------------------------- embeddings
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_ids,
                                                        const int64_t* __restrict__ type_ids, const T* __restrict__ word,
                                                        const T* __restrict__ pos, const T* __restrict__ type,
                                                        T* __restrict__ out, int n_tokens, int seq_len, int d, int vocab,
                                                        int max_pos, int n_types) {
  const int lane = threadIdx.x & 63;
  const int n = imt_xcd_block(blockIdx.x, gridDim.x) * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (n >= n_tokens) return;
  int64_t wi = ids[n];
  int64_t pi = pos_ids ? pos_ids[n] : (int64_t)(n % seq_len);
  int64_t ti = type_ids ? type_ids[n] : 0;
  // clamp like a defensive gather: out-of-range ids would fault in the reference; here they read row 0
  if (wi < 0 || wi >= vocab) wi = 0;
  if (pi < 0 || pi >= max_pos) pi = 0;
  if (ti < 0 || ti >= n_types) ti = 0;
  const T* wr = word + wi * d;
  const T* pr = pos + pi * d;
  const T* tr = type + ti * d;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 v = Vec4<T>::load(wr + c) + Vec4<T>::load(pr + c);
    v += Vec4<T>::load(tr + c);
    Vec4<T>::store(out + (int64_t)n * d + c, v);
  }
}

// scatter-add: one block walks TOKB tokens; thread owns columns tid, tid+256, ... (a wave's atomics of one
// token row are 256 contiguous bytes -- the shape the memory-side atomic units run at full rate).
// type-table rows (2 languages) are accumulated in registers and flushed once per block.
template <typename T, int NT_REG>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ pos_ids,
                                                        const int64_t* __restrict__ type_ids, const T* __restrict__ dsum,
                                                        float* __restrict__ dword, float* __restrict__ dpos,
                                                        float* __restrict__ dtype_tab, int n_tokens, int seq_len, int d,
                                                        int64_t pad_id, int tokb) {
  // Token set of this workgroup.  Default positions (pos = n % seq_len, tokens laid out [sentences][seq_len]): the tokb
  // tokens are the SAME position of tokb consecutive sentences, so the position-table gradient is summed in a register
  // and costs one atomic per column and workgroup instead of one per token (the kernel runs at the atomic rate of the
  // memory side: two atomics per element before, ~1.06 now).  Explicit position ids (MASS): tokb consecutive tokens.
  const bool posmajor = (pos_ids == nullptr) && (n_tokens % seq_len == 0);
  const int nsent = n_tokens / seq_len;
  int first, stride, count, p_own = 0;
  if (posmajor) {
    p_own = blockIdx.x % seq_len;
    const int s0 = (blockIdx.x / seq_len) * tokb;
    first = s0 * seq_len + p_own; stride = seq_len; count = min(tokb, nsent - s0);
  } else {
    first = imt_xcd_block(blockIdx.x, gridDim.x) * tokb; stride = 1; count = min(tokb, n_tokens - first);
  }
  if (count <= 0) return;
  for (int c = threadIdx.x; c < d; c += 256) {
    float tacc[NT_REG], pacc = 0.f;
#pragma unroll
    for (int k = 0; k < NT_REG; ++k) tacc[k] = 0.f;
    // four tokens per trip, their loads requested together and unconditionally (clamped)
    for (int kb = 0; kb < count; kb += 4) {
      float vb[4];
      int64_t wb[4], pb[4], tb[4];
      const int64_t* pp = pos_ids ? pos_ids : ids;    // a valid stand-in address; the value is replaced below
      const int64_t* tp = type_ids ? type_ids : ids;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int n = first + min(kb + k, count - 1) * stride;
        vb[k] = to_f32<T>(dsum[(int64_t)n * d + c]);
        wb[k] = ids[n];
        pb[k] = pp[n];
        tb[k] = tp[n];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (kb + k >= count) break;
        const int n = first + (kb + k) * stride;
        const float v = vb[k];
        const int64_t wi = wb[k];
        const int64_t ti = type_ids ? tb[k] : 0;
        if (wi != pad_id) atomicAdd(dword + wi * d + c, v);
        if (posmajor) pacc += v;
        else atomicAdd(dpos + (pos_ids ? pb[k] : (int64_t)(n % seq_len)) * d + c, v);
        bool hit = false;
#pragma unroll
        for (int q = 0; q < NT_REG; ++q)
          if (ti == q) { tacc[q] += v; hit = true; }
        if (!hit) atomicAdd(dtype_tab + ti * d + c, v);
      }
    }
    if (posmajor) atomicAdd(dpos + (int64_t)p_own * d + c, pacc);
#pragma unroll
    for (int k = 0; k < NT_REG; ++k)
      if (tacc[k] != 0.f) atomicAdd(dtype_tab + (int64_t)k * d + c, tacc[k]);
  }
}

// ------------------------------------------------------------------------------------------- column sums
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t ldx, int M, int N,
                                                     float* __restrict__ out, int rows_per_block,
                                                     const float* __restrict__ scale_dev) {
  __shared__ f32x4 red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < N) {
    const bool full = (c + 3 < N);
    for (int r = r0 + w; r < r1; r += 4) {
      if (full) acc += Vec4<T>::load(X + (int64_t)r * ldx + c);
      else
        for (int e = 0; e < 4 && c + e < N; ++e) acc[e] += to_f32<T>(X[(int64_t)r * ldx + c + e]);
    }
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && c < N) {
    f32x4 s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    if (scale_dev) s *= scale_dev[0];
    for (int e = 0; e < 4 && c + e < N; ++e) atomicAdd(out + c + e, s[e]);
  }
}

// ------------------------------------------------------------------------------------------- row gather / scatter
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ x, int64_t ldx, const int32_t* __restrict__ idx,
                                                          T* __restrict__ out, int64_t ldo, int n_sel, int d, int scatter) {
  const int lane = threadIdx.x & 63;
  const int r = imt_xcd_block(blockIdx.x, gridDim.x) * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (r >= n_sel) return;
  const int64_t src = scatter ? (int64_t)r : (int64_t)idx[r];
  const int64_t dst = scatter ? (int64_t)idx[r] : (int64_t)r;
  for (int c = lane * 4; c < d; c += 256) {
    typedef typename Vec4<T>::type raw_t;
    *reinterpret_cast<raw_t*>(out + dst * ldo + c) = *reinterpret_cast<const raw_t*>(x + src * ldx + c);
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) Vec4<bf16_t>::store(dst + i, Vec4<float>::load(src + i));
    else
      for (int64_t e = i; e < n; ++e) dst[e] = (bf16_t)src[e];
  }
}

// out = sigmoid(gate + 1e-7) * a + (1 - sigmoid(gate + 1e-7)) * b   (src/image_model.py:217-219, :364-366)
template <typename T>
__global__ __launch_bounds__(256) void gated_mix_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                        const T* __restrict__ gate, T* __restrict__ out, int64_t rows, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (r >= rows) return;
  for (int c = lane * 4; c < d; c += 256) {
    const f32x4 av = Vec4<T>::load(a + r * d + c), bv = Vec4<T>::load(b + r * d + c), gv = Vec4<T>::load(gate + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float s = 1.0f / (1.0f + __expf(-(gv[e] + 1e-7f)));
      o[e] = s * av[e] + (1.0f - s) * bv[e];
    }
    Vec4<T>::store(out + r * d + c, o);
  }
}

// out[r, :] = dropout(x[r, :] + add[r % period, :])   -- the two dropouts and the "+ location_embedding" of the image head
// (src/image_model.py:37-41,77-78).  TI -> TO conversion on the way (fp32 region features enter a bf16 model).  The same
// call with add == nullptr and the forward's seed is the backward of the dropout (element index = r * d + c both times).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void add_rows_dropout_kernel(const TI* __restrict__ x, TO* __restrict__ out,
                                                               const TO* __restrict__ add, int64_t rows, int d, int period,
                                                               uint32_t thresh, float inv_keep, uint64_t seed) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6);
  if (r >= rows) return;
  const TO* ar = add ? add + (int64_t)(r % period) * d : nullptr;
  for (int c = lane * 4; c < d; c += 256) {
    f32x4 v = Vec4<TI>::load(x + r * d + c);
    if (ar) v += Vec4<TO>::load(ar + c);
    if (thresh) dropout_apply4(v, seed, (uint64_t)r * d + c, thresh, inv_keep);
    Vec4<TO>::store(out + r * d + c, v);
  }
}

template <typename T, int NCH>
int ln_fwd_launch(int mode, const LnSrc<T>& src, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int rows,
                  int d, float eps, float p, uint64_t seed, hipStream_t st) {
  static const char* const kinds[3] = {"layernorm_fwd", "add_layernorm_fwd", "embed_ln_fwd"};
  ImtProfScope prof(kinds[mode], 0.0, (mode == 0 ? 2.0 : mode == 1 ? 4.0 : 5.0) * rows * d * sizeof(T), st);
  const dim3 grid(imt_cdiv(rows, ROWS_PER_BLOCK));
  const uint32_t th = dropout_thresh(p);
  const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
  if (mode == 0)
    hipLaunchKernelGGL((ln_fwd_kernel<T, NCH, 0>), grid, dim3(256), 0, st, src, (const T*)gamma, (const T*)beta, (T*)y, mean, rstd, rows, d, eps, th, ik, seed);
  else if (mode == 1)
    hipLaunchKernelGGL((ln_fwd_kernel<T, NCH, 1>), grid, dim3(256), 0, st, src, (const T*)gamma, (const T*)beta, (T*)y, mean, rstd, rows, d, eps, th, ik, seed);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<T, NCH, 2>), grid, dim3(256), 0, st, src, (const T*)gamma, (const T*)beta, (T*)y, mean, rstd, rows, d, eps, th, ik, seed);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

template <typename T, int NCH>
int ln_bwd_launch(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd, void* dx,
                  float* dgamma, float* dbeta, int rows, int d, float yp, uint64_t yseed, void* dx_drop, float dxp,
                  uint64_t dxseed, float* partial, hipStream_t st) {
  // Without a partial-sum workspace every workgroup ends with one fp32 atomic per column straight into dgamma / dbeta,
  // and that tail is what costs, so the grid stays at ~256 workgroups and the row phase gets its memory-level
  // parallelism from WPB waves per workgroup (sweeps: profiles/r01_ln_bwd_sweep.txt).  With the workspace (the stack
  // backward) the atomics spread over 32 copies and 512 workgroups are faster (C1 step 7.43 -> 7.35 ms).
  int wpb = 8;
  if (const char* e = getenv("IMT_LN_WPB")) wpb = atoi(e);  // tuning hook: 4 | 8 | 16
  int nblk = partial ? 512 : 256;
  if (const char* e = getenv("IMT_LN_BLOCKS")) nblk = atoi(e) > 0 ? atoi(e) : nblk;
  int rpw = imt_cdiv(rows, nblk * wpb);
  if (rpw < 1) rpw = 1;
  const int blocks = imt_cdiv(rows, rpw * wpb);
  ImtProfScope prof("layernorm_bwd", 0.0, (dx_drop ? 4.0 : 3.0) * rows * d * sizeof(T), st);
#define IMT_LN_BWD_LAUNCH(W)                                                                                          \
  hipLaunchKernelGGL((ln_bwd_kernel<T, NCH, W>), dim3(blocks), dim3(W * 64), 0, st, (const T*)dy, (const T*)x,        \
                     (const T*)gamma, mean, rstd, (T*)dx, dgamma, dbeta, rows, d, rpw, dropout_thresh(yp),            \
                     yp > 0.f ? 1.f / (1.f - yp) : 1.f, yseed, (T*)dx_drop, dropout_thresh(dxp),                      \
                     dxp > 0.f ? 1.f / (1.f - dxp) : 1.f, dxseed, partial, trace)
  ImtTrace tr("ln_bwd", blocks, st);  // phases: loads issued | first row reduced | first row stored | (unused) | other rows | tail
  unsigned long long* trace = tr.dev;
  if (wpb == 4) IMT_LN_BWD_LAUNCH(4);
  else if (wpb == 8) IMT_LN_BWD_LAUNCH(8);
  else IMT_LN_BWD_LAUNCH(16);
#undef IMT_LN_BWD_LAUNCH
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

#define IMT_DISPATCH_NCH(FN, T, d, ...)                                          \
  do {                                                                           \
    const int nch__ = imt_cdiv(d, 256);                                          \
    if (nch__ <= 1) return FN<T, 1>(__VA_ARGS__);                                \
    if (nch__ <= 2) return FN<T, 2>(__VA_ARGS__);                                \
    if (nch__ <= 3) return FN<T, 3>(__VA_ARGS__);                                \
    if (nch__ <= 4) return FN<T, 4>(__VA_ARGS__);                                \
    imt_set_error("layernorm: d=%d > 1024 unsupported", d);                      \
    return IMT_ERR_UNSUPPORTED;                                                  \
  } while (0)

}  // namespace

template <typename T>
int ln_fwd_dispatch(int mode, const LnSrc<T>& src, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int rows,
                    int d, float eps, float p, uint64_t seed, hipStream_t st) {
  IMT_DISPATCH_NCH(ln_fwd_launch, T, d, mode, src, gamma, beta, y, mean, rstd, rows, d, eps, p, seed, st);
}

extern "C" int imt_layernorm_fwd(int dtype, const void* x, const void* gamma, const void* beta, void* y, float* mean,
                                 float* rstd, int rows, int d, float eps, float dropout_p, uint64_t dropout_seed,
                                 void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "layernorm_fwd: bad dtype");
  IMT_CHECK_ARG(d > 0 && d % 4 == 0, "layernorm_fwd: d must be a positive multiple of 4");
  if (rows <= 0) return IMT_OK;
  IMT_CHECK_ARG(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IMT_F32) { LnSrc<float> s{}; s.x = (const float*)x; return ln_fwd_dispatch<float>(0, s, gamma, beta, y, mean, rstd, rows, d, eps, dropout_p, dropout_seed, st); }
  LnSrc<bf16_t> s{}; s.x = (const bf16_t*)x;
  return ln_fwd_dispatch<bf16_t>(0, s, gamma, beta, y, mean, rstd, rows, d, eps, dropout_p, dropout_seed, st);
}

extern "C" int imt_add_layernorm_fwd(int dtype, const void* x, const void* resid, const void* gamma, const void* beta, void* sum_out,
                                     void* y, float* mean, float* rstd, int rows, int d, float eps, float dropout_p,
                                     uint64_t dropout_seed, void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "add_layernorm_fwd: bad dtype");
  IMT_CHECK_ARG(d > 0 && d % 4 == 0, "add_layernorm_fwd: d must be a positive multiple of 4");
  if (rows <= 0) return IMT_OK;
  IMT_CHECK_ARG(x && resid && gamma && beta && sum_out && y && mean && rstd, "add_layernorm_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IMT_F32) {
    LnSrc<float> s{}; s.x = (const float*)x; s.resid = (const float*)resid; s.sum_out = (float*)sum_out;
    return ln_fwd_dispatch<float>(1, s, gamma, beta, y, mean, rstd, rows, d, eps, dropout_p, dropout_seed, st);
  }
  LnSrc<bf16_t> s{}; s.x = (const bf16_t*)x; s.resid = (const bf16_t*)resid; s.sum_out = (bf16_t*)sum_out;
  return ln_fwd_dispatch<bf16_t>(1, s, gamma, beta, y, mean, rstd, rows, d, eps, dropout_p, dropout_seed, st);
}

extern "C" int imt_embed_ln_fwd(int dtype, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids, const void* word,
                                const void* pos, const void* type, const void* gamma, const void* beta, void* sum_out, void* y,
                                float* mean, float* rstd, int n_tokens, int seq_len, int d, int vocab, int max_pos, int n_types,
                                float eps, float dropout_p, uint64_t dropout_seed, void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "embed_ln_fwd: bad dtype");
  IMT_CHECK_ARG(d > 0 && d % 4 == 0 && seq_len > 0, "embed_ln_fwd: bad dims");
  if (n_tokens <= 0) return IMT_OK;
  IMT_CHECK_ARG(ids && word && pos && type && gamma && beta && sum_out && y && mean && rstd, "embed_ln_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IMT_F32) {
    LnSrc<float> s{}; s.ids = ids; s.pos_ids = pos_ids; s.type_ids = type_ids; s.word = (const float*)word; s.pos = (const float*)pos;
    s.type = (const float*)type; s.sum_out = (float*)sum_out; s.seq_len = seq_len; s.vocab = vocab; s.max_pos = max_pos; s.n_types = n_types;
    return ln_fwd_dispatch<float>(2, s, gamma, beta, y, mean, rstd, n_tokens, d, eps, dropout_p, dropout_seed, st);
  }
  LnSrc<bf16_t> s{}; s.ids = ids; s.pos_ids = pos_ids; s.type_ids = type_ids; s.word = (const bf16_t*)word; s.pos = (const bf16_t*)pos;
  s.type = (const bf16_t*)type; s.sum_out = (bf16_t*)sum_out; s.seq_len = seq_len; s.vocab = vocab; s.max_pos = max_pos; s.n_types = n_types;
  return ln_fwd_dispatch<bf16_t>(2, s, gamma, beta, y, mean, rstd, n_tokens, d, eps, dropout_p, dropout_seed, st);
}

// grads[g_off[i] + c] += sum_k partials[i][k][0][c], grads[b_off[i] + c] += sum_k partials[i][k][1][c]   (k over the copies)
constexpr int LN_MAX_SITES = 224;
struct LnReduceArgs { int n, d; int64_t g_off[LN_MAX_SITES], b_off[LN_MAX_SITES]; };
__global__ __launch_bounds__(256) void ln_partial_reduce_kernel(const float* __restrict__ partials, float* __restrict__ grads,
                                                                LnReduceArgs a) {
  const int site = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 2 * a.d || a.g_off[site] < 0) return;  // a negative offset marks an unused slot of the caller's layout
  const float* p = partials + (size_t)site * IMT_LN_PARTIAL_COPIES * 2 * a.d + c;  // [copies][2*d]
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < IMT_LN_PARTIAL_COPIES; ++k) sum += p[(size_t)k * 2 * a.d];
  float* g = grads + (c < a.d ? a.g_off[site] + c : a.b_off[site] + (c - a.d));
  *g += sum;
}

extern "C" int imt_ln_partial_reduce(const float* partials, int n_sites, int d, const int64_t* host_dgamma_off,
                                     const int64_t* host_dbeta_off, float* grads, void* stream) {
  if (n_sites <= 0) return IMT_OK;
  IMT_CHECK_ARG(partials && grads && host_dgamma_off && host_dbeta_off && d > 0, "ln_partial_reduce: bad args");
  IMT_CHECK_ARG(n_sites <= LN_MAX_SITES, "ln_partial_reduce: at most %d sites per call", LN_MAX_SITES);
  LnReduceArgs a;
  a.n = n_sites; a.d = d;
  for (int i = 0; i < n_sites; ++i) { a.g_off[i] = host_dgamma_off[i]; a.b_off[i] = host_dbeta_off[i]; }
  hipStream_t st = (hipStream_t)stream;
  ImtProfScope prof("ln_partial_reduce", 0.0, (double)n_sites * (2.0 * IMT_LN_PARTIAL_COPIES + 2.0) * d * 4.0, st);
  hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3(imt_cdiv(2 * d, 256), n_sites), dim3(256), 0, st, partials, grads, a);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_layernorm_bwd(int dtype, const void* dy, const void* x, const void* gamma, const float* mean,
                                 const float* rstd, void* dx, float* dgamma, float* dbeta, int rows, int d,
                                 float y_dropout_p, uint64_t y_dropout_seed, void* dx_drop, float dx_dropout_p,
                                 uint64_t dx_dropout_seed, float* partial_ws, void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "layernorm_bwd: bad dtype");
  IMT_CHECK_ARG(d > 0 && d % 4 == 0, "layernorm_bwd: d must be a positive multiple of 4");
  if (rows <= 0) return IMT_OK;
  IMT_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "layernorm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == IMT_F32)
    IMT_DISPATCH_NCH(ln_bwd_launch, float, d, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, d, y_dropout_p,
                     y_dropout_seed, dx_drop, dx_dropout_p, dx_dropout_seed, partial_ws, st);
  IMT_DISPATCH_NCH(ln_bwd_launch, bf16_t, d, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, d, y_dropout_p,
                   y_dropout_seed, dx_drop, dx_dropout_p, dx_dropout_seed, partial_ws, st);
}

extern "C" int imt_embed_fwd(int dtype, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                             const void* word, const void* pos, const void* type, void* out, int n_tokens, int seq_len,
                             int d, int vocab, int max_pos, int n_types, void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "embed_fwd: bad dtype");
  IMT_CHECK_ARG(d > 0 && d % 4 == 0 && seq_len > 0, "embed_fwd: bad dims");
  if (n_tokens <= 0) return IMT_OK;
  IMT_CHECK_ARG(ids && word && pos && type && out, "embed_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(imt_cdiv(n_tokens, ROWS_PER_BLOCK));
  ImtProfScope prof("embed_fwd", 0.0, 4.0 * n_tokens * d * (dtype == IMT_BF16 ? 2 : 4), st);
  if (dtype == IMT_F32)
    hipLaunchKernelGGL(embed_fwd_kernel<float>, grid, dim3(256), 0, st, ids, pos_ids, type_ids, (const float*)word,
                       (const float*)pos, (const float*)type, (float*)out, n_tokens, seq_len, d, vocab, max_pos, n_types);
  else
    hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, ids, pos_ids, type_ids, (const bf16_t*)word,
                       (const bf16_t*)pos, (const bf16_t*)type, (bf16_t*)out, n_tokens, seq_len, d, vocab, max_pos, n_types);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_embed_bwd(int dtype, const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                             const void* dsum, float* dword, float* dpos, float* dtype_tab, int n_tokens, int seq_len,
                             int d, int64_t pad_id, void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "embed_bwd: bad dtype");
  IMT_CHECK_ARG(d > 0 && seq_len > 0, "embed_bwd: bad dims");
  if (n_tokens <= 0) return IMT_OK;
  IMT_CHECK_ARG(ids && dsum && dword && dpos && dtype_tab, "embed_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  static const int tokb_env = getenv("IMT_EMBED_TOKB") ? atoi(getenv("IMT_EMBED_TOKB")) : 0;  // tuning only
  const int tokb = tokb_env > 0 ? tokb_env : 16;
  const bool posmajor = (pos_ids == nullptr) && (n_tokens % seq_len == 0);  // must match the kernel's own test
  dim3 grid(posmajor ? seq_len * imt_cdiv(n_tokens / seq_len, tokb) : imt_cdiv(n_tokens, tokb));
  ImtProfScope prof("embed_bwd", 0.0, (double)n_tokens * d * ((dtype == IMT_BF16 ? 2 : 4) + 16.0), st);
  if (dtype == IMT_F32)
    hipLaunchKernelGGL((embed_bwd_kernel<float, 4>), grid, dim3(256), 0, st, ids, pos_ids, type_ids, (const float*)dsum,
                       dword, dpos, dtype_tab, n_tokens, seq_len, d, pad_id, tokb);
  else
    hipLaunchKernelGGL((embed_bwd_kernel<bf16_t, 4>), grid, dim3(256), 0, st, ids, pos_ids, type_ids, (const bf16_t*)dsum,
                       dword, dpos, dtype_tab, n_tokens, seq_len, d, pad_id, tokb);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_colsum(int dtype, const void* X, int64_t ldx, int M, int N, float* out, const float* scale_dev,
                          void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "colsum: bad dtype");
  if (M <= 0 || N <= 0) return IMT_OK;
  IMT_CHECK_ARG(X && out && ldx % 4 == 0, "colsum: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int rpb = 64;
  dim3 grid(imt_cdiv(N, 256), imt_cdiv(M, rpb));
  ImtProfScope prof("colsum", 0.0, (double)M * N * (dtype == IMT_BF16 ? 2 : 4), st);
  if (dtype == IMT_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float*)X, ldx, M, N, out, rpb, scale_dev);
  else
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)X, ldx, M, N, out, rpb, scale_dev);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

static int gather_scatter(int dtype, const void* x, int64_t ldx, const int32_t* idx, void* out, int64_t ldo, int n_sel,
                          int d, int scatter, void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "gather_rows: bad dtype");
  IMT_CHECK_ARG(d > 0 && d % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, "gather_rows: d/ld must be multiples of 4");
  if (n_sel <= 0) return IMT_OK;
  IMT_CHECK_ARG(x && idx && out, "gather_rows: null pointer");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(imt_cdiv(n_sel, ROWS_PER_BLOCK));
  ImtProfScope prof(scatter ? "scatter_rows" : "gather_rows", 0.0, 2.0 * n_sel * d * (dtype == IMT_BF16 ? 2 : 4), st);
  if (dtype == IMT_F32)
    hipLaunchKernelGGL(gather_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, idx, (float*)out, ldo, n_sel, d, scatter);
  else
    hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, idx, (bf16_t*)out, ldo, n_sel, d, scatter);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_gather_rows(int dtype, const void* x, int64_t ldx, const int32_t* idx, void* out, int64_t ldo,
                               int n_sel, int d, void* stream) {
  return gather_scatter(dtype, x, ldx, idx, out, ldo, n_sel, d, 0, stream);
}
extern "C" int imt_scatter_rows(int dtype, const void* dout, int64_t ldo, const int32_t* idx, void* dx, int64_t ldx,
                                int n_sel, int d, void* stream) {
  return gather_scatter(dtype, dout, ldo, idx, dx, ldx, n_sel, d, 1, stream);
}

extern "C" int imt_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (n <= 0) return IMT_OK;
  IMT_CHECK_ARG(src && dst, "cast: null pointer");
  int blocks = imt_cdiv(n, 1024);
  if (blocks > 2048) blocks = 2048;
  ImtProfScope prof("cast_f32_to_bf16", 0.0, 6.0 * n, (hipStream_t)stream);
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_add_rows_dropout(int in_dtype, const void* x, int out_dtype, void* out, const void* add, int64_t rows,
                                    int d, int period, float dropout_p, uint64_t dropout_seed, void* stream) {
  IMT_CHECK_ARG((in_dtype == IMT_F32 || in_dtype == IMT_BF16) && (out_dtype == IMT_F32 || out_dtype == IMT_BF16), "add_rows_dropout: bad dtype");
  IMT_CHECK_ARG(d > 0 && d % 4 == 0, "add_rows_dropout: d must be a multiple of 4");
  IMT_CHECK_ARG(!add || period > 0, "add_rows_dropout: period must be positive");
  IMT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "add_rows_dropout: dropout_p outside [0, 1)");
  if (rows <= 0) return IMT_OK;
  IMT_CHECK_ARG(x && out, "add_rows_dropout: null pointer");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(imt_cdiv(rows, ROWS_PER_BLOCK));
  const uint32_t th = dropout_thresh(dropout_p);
  const float ik = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
  ImtProfScope prof("add_rows_dropout", 0.0, (double)rows * d * ((in_dtype == IMT_BF16 ? 2 : 4) + (out_dtype == IMT_BF16 ? 2 : 4)), st);
#define IMT_ARD(TI, TO) hipLaunchKernelGGL((add_rows_dropout_kernel<TI, TO>), grid, dim3(256), 0, st, (const TI*)x, (TO*)out, (const TO*)add, rows, d, period, th, ik, dropout_seed)
  if (in_dtype == IMT_F32 && out_dtype == IMT_F32) IMT_ARD(float, float);
  else if (in_dtype == IMT_F32) IMT_ARD(float, bf16_t);
  else if (out_dtype == IMT_F32) IMT_ARD(bf16_t, float);
  else IMT_ARD(bf16_t, bf16_t);
#undef IMT_ARD
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_gated_mix(int dtype, const void* a, const void* b, const void* gate, void* out, int64_t rows, int d,
                             void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "gated_mix: bad dtype");
  IMT_CHECK_ARG(d > 0 && d % 4 == 0, "gated_mix: d must be a multiple of 4");
  if (rows <= 0) return IMT_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(imt_cdiv(rows, ROWS_PER_BLOCK));
  if (dtype == IMT_F32)
    hipLaunchKernelGGL(gated_mix_kernel<float>, grid, dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)gate, (float*)out, rows, d);
  else
    hipLaunchKernelGGL(gated_mix_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)gate, (bf16_t*)out, rows, d);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}
