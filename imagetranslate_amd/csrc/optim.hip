// Optimizer step over the FLAT parameter / gradient / moment buffers (one launch each):
//   global grad-norm (clip_grad_norm_, src/train_image_mt.py:291) and Adam with fp32 master weights
//   (torch.optim.Adam as subclassed by AdamInverseSqrtWithWarmup, src/utils.py:105-156), writing the bf16
//   shadow copy the MFMA kernels read and zeroing the gradients for the next step in the same pass.
// HBM-bound: 4 fp32 streams read, 3 written (+1 bf16) per element.
#include <stdlib.h>
#include "common.hpp"

namespace {

// Deterministic: every workgroup writes ONE partial (fixed element->thread mapping, fixed in-block order) and a second
// one-block kernel adds the partials in a fixed tree order.  (A float atomicAdd per workgroup made the global norm --
// hence the clip coefficient, hence every parameter -- differ in the last bit between data-parallel ranks that hold
// identical all-reduced gradients: replicas must stay bit-identical without ever re-synchronising parameters.)
constexpr int SUMSQ_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 1024;
  const int64_t n4 = n & ~(int64_t)3;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n4; i += stride) {
    const f32x4 v = Vec4<float>::load(g + i);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] += (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamP {
  float max_norm, grad_scale, lr, beta1, beta2, eps, bc1, bc2_sqrt;
  int zero_grad;
};

IMT_DEVICE float adam_one(float& p, float g, float& m, float& v, const AdamP& a, float coef) {
  g *= coef;
  m = a.beta1 * m + (1.f - a.beta1) * g;
  v = a.beta2 * v + (1.f - a.beta2) * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p -= (a.lr / a.bc1) * (m / denom);
  return p;
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, bf16_t* __restrict__ pb, int64_t n,
                                                        const float* __restrict__ sumsq, AdamP a) {
  float coef = a.grad_scale;
  if (sumsq && a.max_norm > 0.f) {
    const float total_norm = a.grad_scale * sqrtf(sumsq[0]);
    const float c = a.max_norm / (total_norm + 1e-6f);
    coef *= (c < 1.f ? c : 1.f);
  }
  const int64_t stride = (int64_t)gridDim.x * 1024;
  const int64_t n4 = n & ~(int64_t)3;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n4; i += stride) {
    f32x4 pv = Vec4<float>::load(p + i), gv = Vec4<float>::load(g + i), mv = Vec4<float>::load(m + i), vv = Vec4<float>::load(v + i);
    // untouched parameters (zero gradient and still-zero moments: the other language's output layer, vocabulary rows
    // no batch has contained yet) get an exactly-zero update: nothing to write back (18 of 34 bytes per element saved)
    bool idle = true;
#pragma unroll
    for (int e = 0; e < 4; ++e) idle = idle && gv[e] == 0.f && mv[e] == 0.f && vv[e] == 0.f;
    if (idle) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pv[e], me = mv[e], ve = vv[e];
      adam_one(pe, gv[e], me, ve, a, coef);
      pv[e] = pe; mv[e] = me; vv[e] = ve;
    }
    Vec4<float>::store(p + i, pv);
    Vec4<float>::store(m + i, mv);
    Vec4<float>::store(v + i, vv);
    if (pb) Vec4<bf16_t>::store(pb + i, pv);
    if (a.zero_grad) Vec4<float>::store(g + i, f32x4{0.f, 0.f, 0.f, 0.f});
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 + threadIdx.x; i < n; i += 256) {
      adam_one(p[i], g[i], m[i], v[i], a, coef);
      if (pb) pb[i] = (bf16_t)p[i];
      if (a.zero_grad) g[i] = 0.f;
    }
}

// clip_grad_norm_ alone (in place): the micro-steps of a gradient-accumulation window that do not end in an optimizer
// step (src/train_image_mt.py:291 clips the accumulated gradient after EVERY backward, :292-295 steps every `accum`)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ sumsq,
                                                         float max_norm, float grad_scale) {
  float coef = grad_scale;
  if (sumsq && max_norm > 0.f) {
    const float total_norm = grad_scale * sqrtf(sumsq[0]);
    const float c = max_norm / (total_norm + 1e-6f);
    coef *= (c < 1.f ? c : 1.f);
  }
  if (coef == 1.f) return;
  const int64_t stride = (int64_t)gridDim.x * 1024;
  const int64_t n4 = n & ~(int64_t)3;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n4; i += stride) {
    f32x4 gv = Vec4<float>::load(g + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) gv[e] *= coef;
    Vec4<float>::store(g + i, gv);
  }
  if (blockIdx.x == 0)
    for (int64_t i = n4 + threadIdx.x; i < n; i += 256) g[i] *= coef;
}

}  // namespace

extern "C" int imt_clip_scale(float* g, int64_t n, const float* sumsq, float max_norm, float grad_scale, void* stream) {
  if (n <= 0) return IMT_OK;
  IMT_CHECK_ARG(g, "clip_scale: null pointer");
  IMT_CHECK_ARG(((uintptr_t)g & 15) == 0, "clip_scale: 16-B alignment");
  int blocks = imt_cdiv(n, 4096);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  ImtProfScope prof("clip_scale", 0.0, 8.0 * n, (hipStream_t)stream);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, sumsq, max_norm, grad_scale);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_sumsq(const float* g, int64_t n, float* out, float* partial_ws, void* stream) {
  if (n <= 0) return IMT_OK;
  IMT_CHECK_ARG(g && out && partial_ws, "sumsq: null pointer");
  IMT_CHECK_ARG(((uintptr_t)g & 15) == 0, "sumsq: 16-B alignment");
  int blocks = imt_cdiv(n, 4096);
  if (blocks > SUMSQ_MAX_BLOCKS) blocks = SUMSQ_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  ImtProfScope prof("grad_sumsq", 0.0, 4.0 * n, (hipStream_t)stream);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, partial_ws);
  IMT_CHECK_LAUNCH();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partial_ws, blocks, out);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_clip_adam(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const float* sumsq,
                             float max_norm, float grad_scale, float lr, float beta1, float beta2, float eps, int64_t step,
                             int zero_grad, void* stream) {
  if (n <= 0) return IMT_OK;
  IMT_CHECK_ARG(p && g && m && v, "clip_adam: null pointer");
  IMT_CHECK_ARG(step >= 1, "clip_adam: step must be >= 1");
  IMT_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)p_bf16) & 15) == 0, "clip_adam: 16-B alignment");
  AdamP a;
  a.max_norm = max_norm; a.grad_scale = grad_scale; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.zero_grad = zero_grad;
  // grid: enough workgroups to saturate HBM when the update runs alone (2048 x 256 threads: every wave slot of the chip).
  // IMT_ADAM_BLOCKS (tuning): a smaller grid leaves wave slots to the kernels of the next forward when the update runs on a
  // side stream beside them.
  static const int cap = getenv("IMT_ADAM_BLOCKS") ? atoi(getenv("IMT_ADAM_BLOCKS")) : 2048;
  int blocks = imt_cdiv(n, 2048);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  ImtProfScope prof("clip_adam", 0.0, (p_bf16 ? 34.0 : 32.0) * n, (hipStream_t)stream);
  hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16, n, sumsq, a);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}
