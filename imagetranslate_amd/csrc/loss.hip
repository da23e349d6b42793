// Vocabulary-row kernels: log-softmax (src/seq2seq.py:179-180), label-smoothed NLL (src/loss.py:10-27) and the
// fused cross-entropy forward+backward used by the training fast path.  One 256-thread workgroup per row;
// a row (V = 30k..60k logits) is streamed with 8/16-byte loads and reduced with wave + LDS reductions.
#include <stdlib.h>
#include "common.hpp"

namespace {

IMT_DEVICE float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
IMT_DEVICE float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// row max and sum(exp(z - max)), plus sum(z); V % 4 == 0 fast path, scalar tail otherwise
template <typename T>
IMT_DEVICE void row_stats(const T* z, int V, float* red, float& mx, float& se, float& sz) {
  float m = -INFINITY;
  const int V4 = V & ~3;
  for (int c = threadIdx.x * 4; c < V4; c += 1024) {
    const f32x4 v = Vec4<T>::load(z + c);
    m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  for (int c = V4 + threadIdx.x; c < V; c += 256) m = fmaxf(m, to_f32<T>(z[c]));
  mx = block_max(m, red);
  float s = 0.f, t = 0.f;
  for (int c = threadIdx.x * 4; c < V4; c += 1024) {
    const f32x4 v = Vec4<T>::load(z + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { s += __expf(v[e] - mx); t += v[e]; }
  }
  for (int c = V4 + threadIdx.x; c < V; c += 256) { const float v = to_f32<T>(z[c]); s += __expf(v - mx); t += v; }
  se = block_sum(s, red);
  sz = block_sum(t, red);
}

template <typename T>
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const T* __restrict__ logits, int64_t ld, float* __restrict__ lp,
                                                              int64_t ldlp, float* __restrict__ lse_out, int V) {
  __shared__ float red[4];
  const int64_t row = imt_xcd_block(blockIdx.x, gridDim.x);
  const T* z = logits + row * ld;
  float mx, se, sz;
  row_stats<T>(z, V, red, mx, se, sz);
  const float lse = mx + __logf(se);
  if (threadIdx.x == 0 && lse_out) lse_out[row] = lse;
  float* o = lp + row * ldlp;
  const int V4 = V & ~3;
  for (int c = threadIdx.x * 4; c < V4; c += 1024) {
    f32x4 v = Vec4<T>::load(z + c);
    v -= lse;
    Vec4<float>::store(o + c, v);
  }
  for (int c = V4 + threadIdx.x; c < V; c += 256) o[c] = to_f32<T>(z[c]) - lse;
}

// dlogits = dlp - exp(lp) * sum_v dlp
template <typename TO>
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ dlp, int64_t lddlp,
                                                              const float* __restrict__ lp, int64_t ldlp,
                                                              TO* __restrict__ dlogits, int64_t ld, int V) {
  __shared__ float red[4];
  const int64_t row = imt_xcd_block(blockIdx.x, gridDim.x);
  const float* g = dlp + row * lddlp;
  const float* l = lp + row * ldlp;
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) s += g[c];
  const float gs = block_sum(s, red);
  TO* o = dlogits + row * ld;
  for (int c = threadIdx.x; c < V; c += 256) o[c] = from_f32<TO>(g[c] - __expf(l[c]) * gs);
}

__global__ __launch_bounds__(256) void smoothed_nll_fwd_kernel(const float* __restrict__ lp, int64_t ldlp,
                                                               const int64_t* __restrict__ target, float* __restrict__ loss,
                                                               int V, float eps, int64_t ignore_index) {
  __shared__ float red[4];
  const int64_t row = imt_xcd_block(blockIdx.x, gridDim.x);
  const int64_t t = target[row];
  const float* l = lp + row * ldlp;
  float s = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) s += l[c];
  const float total = block_sum(s, red);
  if (threadIdx.x == 0) {
    float out = 0.f;
    if (t != ignore_index) {
      const float nll = (t >= 0 && t < V) ? -l[t] : 0.f;
      out = (1.f - eps) * nll + (eps / (float)V) * (-total);
    }
    loss[row] = out;
  }
}

__global__ __launch_bounds__(256) void smoothed_nll_bwd_kernel(const float* __restrict__ dloss, const int64_t* __restrict__ target,
                                                               float* __restrict__ dlp, int64_t lddlp, int V, float eps,
                                                               int64_t ignore_index) {
  const int64_t row = imt_xcd_block(blockIdx.x, gridDim.x);
  const int64_t t = target[row];
  const float g = (t == ignore_index) ? 0.f : dloss[row];
  const float base = -g * (eps / (float)V);
  float* o = dlp + row * lddlp;
  for (int c = threadIdx.x; c < V; c += 256) o[c] = base - ((c == t) ? g * (1.f - eps) : 0.f);
}

// fused: loss_row and dlogits (in place over the logits)
template <typename T>
__global__ __launch_bounds__(256) void xent_fused_kernel(T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ target,
                                                         float* __restrict__ loss_rows, int V, float eps, int64_t ignore_index,
                                                         float grad_scale) {
  __shared__ float red[4];
  const int64_t row = imt_xcd_block(blockIdx.x, gridDim.x);
  T* z = logits + row * ld;
  const int64_t t = target[row];
  const bool ignored = (t == ignore_index);
  float mx, se, sz;
  row_stats<T>(z, V, red, mx, se, sz);
  const float lse = mx + __logf(se);
  if (threadIdx.x == 0) {
    float out = 0.f;
    if (!ignored) {
      const float zt = (t >= 0 && t < V) ? to_f32<T>(z[t]) : lse;
      out = (1.f - eps) * (lse - zt) + (eps / (float)V) * ((float)V * lse - sz);
    }
    loss_rows[row] = out;
  }
  __syncthreads();  // z[t] read above before anyone overwrites it
  const float gs = ignored ? 0.f : grad_scale;
  const float sm = eps / (float)V;
  const int V4 = V & ~3;
  for (int c = threadIdx.x * 4; c < V4; c += 1024) {
    f32x4 v = Vec4<T>::load(z + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gs * (__expf(v[e] - lse) - sm - ((c + e == t) ? (1.f - eps) : 0.f));
    Vec4<T>::store(z + c, v);
  }
  for (int c = V4 + threadIdx.x; c < V; c += 256)
    z[c] = from_f32<T>(gs * (__expf(to_f32<T>(z[c]) - lse) - sm - ((c == t) ? (1.f - eps) : 0.f)));
}

// bf16 rows are small enough (V = 30000 -> 60 KB) to park in LDS: the row is read from HBM ONCE (while taking the
// maximum), the exp-sum and the gradient pass read the LDS copy, and dlogits is written once -- 2 x N x V x 2 bytes of
// HBM traffic instead of 4 x (profiles/r01_pmc_traffic.json showed 1.95 GB for the three-read version).
__global__ __launch_bounds__(256) void xent_fused_lds_kernel(bf16_t* __restrict__ logits, int64_t ld,
                                                             const int64_t* __restrict__ target, float* __restrict__ loss_rows,
                                                             int V, float eps, int64_t ignore_index, float grad_scale) {
  extern __shared__ __attribute__((aligned(16))) char row_s[];
  __shared__ float red[4];
  bf16_t* zs = reinterpret_cast<bf16_t*>(row_s);
  const int64_t row = imt_xcd_block(blockIdx.x, gridDim.x);
  bf16_t* z = logits + row * ld;
  const int64_t t = target[row];
  const bool ignored = (t == ignore_index);
  const int V8 = V & ~7;
  float m = -INFINITY, sumz = 0.f;
  // 8 independent 16-byte loads per thread in flight (a plain loop would pay one HBM latency per iteration)
  constexpr int UNR = 8;
  for (int c0 = threadIdx.x * 8; c0 < V8; c0 += 2048 * UNR) {
    bf16x8 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int c = c0 + u * 2048;
      if (c < V8) v[u] = *reinterpret_cast<const bf16x8*>(z + c);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int c = c0 + u * 2048;
      if (c < V8) {
        *reinterpret_cast<bf16x8*>(zs + c) = v[u];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[u][e]; m = fmaxf(m, f); sumz += f; }
      }
    }
  }
  for (int c = V8 + threadIdx.x; c < V; c += 256) { const bf16_t v = z[c]; zs[c] = v; m = fmaxf(m, (float)v); sumz += (float)v; }
  const float mx = block_max(m, red);  // (barrier inside: the LDS copy is complete)
  float s = 0.f;
  for (int c = threadIdx.x * 8; c < V8; c += 2048) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(zs + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += __expf((float)v[e] - mx);
  }
  for (int c = V8 + threadIdx.x; c < V; c += 256) s += __expf((float)zs[c] - mx);
  const float se = block_sum(s, red);
  const float sz = block_sum(sumz, red);
  const float lse = mx + __logf(se);
  if (threadIdx.x == 0) {
    float out = 0.f;
    if (!ignored) {
      const float zt = (t >= 0 && t < V) ? (float)zs[t] : lse;
      out = (1.f - eps) * (lse - zt) + (eps / (float)V) * ((float)V * lse - sz);
    }
    loss_rows[row] = out;
  }
  const float gs = ignored ? 0.f : grad_scale;
  const float sm = eps / (float)V;
  for (int c = threadIdx.x * 8; c < V8; c += 2048) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(zs + c);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(gs * (__expf((float)v[e] - lse) - sm - ((c + e == t) ? (1.f - eps) : 0.f)));
    *reinterpret_cast<bf16x8*>(z + c) = o;
  }
  for (int c = V8 + threadIdx.x; c < V; c += 256)
    z[c] = (bf16_t)(gs * (__expf((float)zs[c] - lse) - sm - ((c == t) ? (1.f - eps) : 0.f)));
}

// out[0] = scale * sum(x[0..n)) in a fixed order (one workgroup; the per-row losses of a step are a few thousand floats)
__global__ __launch_bounds__(1024) void scaled_sum_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) s += x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k];
    out[0] = t * scale;
  }
}

}  // namespace

extern "C" int imt_scaled_sum(const float* x, int n, float scale, float* out, void* stream) {
  IMT_CHECK_ARG(x && out && n >= 0, "scaled_sum: bad args");
  hipLaunchKernelGGL(scaled_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, n, scale, out);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_log_softmax_fwd(int dtype, const void* logits, int64_t ld, float* lp, int64_t ldlp, float* lse, int N,
                                   int V, void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "log_softmax_fwd: bad dtype");
  if (N <= 0) return IMT_OK;
  IMT_CHECK_ARG(logits && lp && V > 0 && ld % 4 == 0 && ldlp % 4 == 0, "log_softmax_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  ImtProfScope prof("log_softmax_fwd", 0.0, (double)N * V * ((dtype == IMT_BF16 ? 2 : 4) + 4.0), st);
  if (dtype == IMT_F32)
    hipLaunchKernelGGL(log_softmax_fwd_kernel<float>, dim3(N), dim3(256), 0, st, (const float*)logits, ld, lp, ldlp, lse, V);
  else
    hipLaunchKernelGGL(log_softmax_fwd_kernel<bf16_t>, dim3(N), dim3(256), 0, st, (const bf16_t*)logits, ld, lp, ldlp, lse, V);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_log_softmax_bwd(const float* dlp, int64_t lddlp, const float* lp, int64_t ldlp, int out_dtype,
                                   void* dlogits, int64_t ld, int N, int V, void* stream) {
  IMT_CHECK_ARG(out_dtype == IMT_F32 || out_dtype == IMT_BF16, "log_softmax_bwd: bad dtype");
  if (N <= 0) return IMT_OK;
  IMT_CHECK_ARG(dlp && lp && dlogits && V > 0, "log_softmax_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == IMT_F32)
    hipLaunchKernelGGL(log_softmax_bwd_kernel<float>, dim3(N), dim3(256), 0, st, dlp, lddlp, lp, ldlp, (float*)dlogits, ld, V);
  else
    hipLaunchKernelGGL(log_softmax_bwd_kernel<bf16_t>, dim3(N), dim3(256), 0, st, dlp, lddlp, lp, ldlp, (bf16_t*)dlogits, ld, V);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_smoothed_nll_fwd(const float* lp, int64_t ldlp, const int64_t* target, float* loss, int N, int V,
                                    float epsilon, int64_t ignore_index, void* stream) {
  if (N <= 0) return IMT_OK;
  IMT_CHECK_ARG(lp && target && loss && V > 0, "smoothed_nll_fwd: bad args");
  hipLaunchKernelGGL(smoothed_nll_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, lp, ldlp, target, loss, V, epsilon, ignore_index);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_smoothed_nll_bwd(const float* dloss, const int64_t* target, float* dlp, int64_t lddlp, int N, int V,
                                    float epsilon, int64_t ignore_index, void* stream) {
  if (N <= 0) return IMT_OK;
  IMT_CHECK_ARG(dloss && target && dlp && V > 0, "smoothed_nll_bwd: bad args");
  hipLaunchKernelGGL(smoothed_nll_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, dloss, target, dlp, lddlp, V, epsilon, ignore_index);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_xent_fused_fwd_bwd(int dtype, void* logits, int64_t ld, const int64_t* target, float* loss_rows, int N,
                                      int V, float epsilon, int64_t ignore_index, float grad_scale, void* stream) {
  IMT_CHECK_ARG(dtype == IMT_F32 || dtype == IMT_BF16, "xent_fused: bad dtype");
  if (N <= 0) return IMT_OK;
  IMT_CHECK_ARG(logits && target && loss_rows && V > 0 && ld % 4 == 0, "xent_fused: bad args");
  hipStream_t st = (hipStream_t)stream;
  ImtProfScope prof("xent_fused", 0.0, 2.0 * N * V * (dtype == IMT_BF16 ? 2 : 4), st);
  const int row_bytes = ((V * 2 + 15) / 16) * 16;
  if (dtype == IMT_F32) {
    hipLaunchKernelGGL(xent_fused_kernel<float>, dim3(N), dim3(256), 0, st, (float*)logits, ld, target, loss_rows, V, epsilon, ignore_index, grad_scale);
  } else if (row_bytes <= 128 * 1024 && ld % 8 == 0 && !getenv("IMT_XENT_NO_LDS")) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(xent_fused_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      attr_set = true;
    }
    hipLaunchKernelGGL(xent_fused_lds_kernel, dim3(N), dim3(256), row_bytes, st, (bf16_t*)logits, ld, target, loss_rows, V, epsilon, ignore_index, grad_scale);
  } else
    hipLaunchKernelGGL(xent_fused_kernel<bf16_t>, dim3(N), dim3(256), 0, st, (bf16_t*)logits, ld, target, loss_rows, V, epsilon, ignore_index, grad_scale);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}
