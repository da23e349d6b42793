// Common device/host helpers for the imagetranslate_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/imt_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define IMT_WAVE 64
#define IMT_DEVICE __device__ __forceinline__

void imt_set_error(const char* fmt, ...);

#define IMT_CHECK_ARG(cond, ...)                       \
  do {                                                  \
    if (!(cond)) {                                      \
      imt_set_error(__VA_ARGS__);                       \
      return IMT_ERR_BAD_ARG;                           \
    }                                                   \
  } while (0)

#define IMT_CHECK_LAUNCH()                                                   \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      imt_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,           \
                    hipGetErrorString(e__));                                 \
      return IMT_ERR_LAUNCH;                                                 \
    }                                                                        \
  } while (0)

static inline int imt_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

bool imt_gemm_ln_ticket_enabled();  // gemm.hip: built with -DIMT_LN_TICKET=1 (imt_gemm's ln_out can normalise in-launch)
// ---------------------------------------------------------------- optional launch profiler (core.hip)
bool imt_prof_enabled();
const char* imt_prof_intern(const char* kind, int M, int N, int K);
void* imt_prof_begin_launch(const char* kind, double flops, double bytes, hipStream_t st);
void imt_prof_end_launch(void* tok, hipStream_t st);
// ---------------------------------------------------------------- phase tracing (tuning aid, IMT_TRACE=<kernel kind>)
// A kernel that takes part gets a device buffer (or nullptr): thread 0 of every workgroup stores wall_clock64() (100 MHz)
// into trace[8 * blockIdx.x + k] at up to 8 phase boundaries; the host then SYNCHRONISES, and prints the mean length of
// each phase and the span first-start -> last-end.  Off (nullptr, no sync) unless IMT_TRACE names the kind.
#include <stdlib.h>
struct ImtTrace {
  static constexpr int MAXB = 4096;
  static constexpr int RING = 64;  // IMT_TRACE_RING=1: no sync per launch; every RING-th traced launch prints the absolute
                                   // first-start / last-end of the last RING launches (the dark time BETWEEN kernels)
  unsigned long long* dev = nullptr;
  const char* kind; int blocks; hipStream_t st;
  struct Slot { const char* kind; int blocks; };
  static Slot* slots() { static Slot s[RING]; return s; }
  static int& count() { static int c = 0; return c; }
  static bool ring() { static const bool r = getenv("IMT_TRACE_RING") != nullptr; return r; }
  ImtTrace(const char* kind_, int blocks_, hipStream_t st_) : kind(kind_), blocks(blocks_), st(st_) {
    static const char* want = getenv("IMT_TRACE");
    if (!want || (strcmp(want, kind_) != 0 && strcmp(want, "all") != 0) || blocks_ > MAXB || blocks_ <= 0) return;
    static unsigned long long* buf = nullptr;
    const size_t slice = (size_t)MAXB * 8;
    if (!buf && hipMalloc(&buf, (ring() ? RING : 1) * slice * sizeof(unsigned long long)) != hipSuccess) return;
    const int slot = ring() ? count() % RING : 0;
    if (ring()) slots()[slot] = Slot{kind_, blocks_};
    dev = buf + slot * slice;
    if (!ring()) (void)hipMemsetAsync(dev, 0, sizeof(unsigned long long) * 8 * blocks_, st);
    else if (slot == 0) (void)hipMemsetAsync(buf, 0, RING * slice * sizeof(unsigned long long), st);  // once per ring: no fill between launches
  }
  ~ImtTrace() {
    if (!dev) return;
    static unsigned long long h[MAXB * 8];
    if (ring()) {
      if (++count() % RING != 0) return;
      (void)hipStreamSynchronize(st);
      unsigned long long* base = dev - (size_t)((count() - 1) % RING) * MAXB * 8;
      unsigned long long origin = 0, prev_end = 0;
      for (int s = 0; s < RING; ++s) {
        const Slot& sl = slots()[s];
        (void)hipMemcpy(h, base + (size_t)s * MAXB * 8, sizeof(unsigned long long) * 8 * sl.blocks, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0, s1 = 0;
        for (int i = 0; i < 8 * sl.blocks; ++i) {
          if (!h[i]) continue;
          if (h[i] < t0) t0 = h[i];
          if (h[i] > t1) t1 = h[i];
          if ((i & 7) == 0 && h[i] > s1) s1 = h[i];
        }
        if (!origin) origin = t0;
        fprintf(stderr, "[ring %2d] %-10s %5d wgs  start %9.2f  last-start +%6.2f  end +%6.2f  dark-before %6.2f us\n", s, sl.kind, sl.blocks,
                (t0 - origin) * 0.01, (s1 - t0) * 0.01, (t1 - t0) * 0.01, prev_end ? ((double)t0 - (double)prev_end) * 0.01 : 0.0);
        prev_end = t1;
      }
      return;
    }
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dev, sizeof(unsigned long long) * 8 * blocks, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int last = 0;
    for (int b = 0; b < blocks; ++b)
      for (int k = 0; k < 8; ++k) {
        const unsigned long long v = h[8 * b + k];
        if (!v) continue;
        if (v < t0) t0 = v;
        if (v > t1) t1 = v;
        if (k > last) last = k;
      }
    for (int b = 0; b < blocks; ++b) {
      unsigned long long prev = h[8 * b];
      for (int k = 1; k <= last; ++k) { const unsigned long long v = h[8 * b + k]; if (v && prev) { ph[k] += (v - prev) * 0.01; prev = v; } }
    }
    fprintf(stderr, "[trace %s] %d workgroups, span %.2f us; mean phases (us):", kind, blocks, (t1 - t0) * 0.01);
    for (int k = 1; k <= last; ++k) fprintf(stderr, " %.2f", ph[k] / blocks);
    fprintf(stderr, "\n");
  }
};
#define IMT_STAMP(trace, k) do { if ((trace) && threadIdx.x == 0) (trace)[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)

struct ImtProfScope {
  void* tok; hipStream_t st;
  ImtProfScope(const char* kind, double flops, double bytes, hipStream_t s)
      : tok(imt_prof_enabled() ? imt_prof_begin_launch(kind, flops, bytes, s) : nullptr), st(s) {}
  ~ImtProfScope() { if (tok) imt_prof_end_launch(tok, st); }
};

// ---------------------------------------------------------------- XCD-affine work distribution
// Workgroup b runs on XCD b % 8 (measured on MI355X, tools/probe_xcc.hip: strict round-robin, block 0 on XCC 0 in
// every launch).  Every kernel therefore maps hardware block id -> logical work item with the SAME bijection
// below: XCD x owns the x-th contiguous eighth of the work (token rows), so what one kernel writes stays in the L2
// of the XCD whose workgroups read it in the next kernel (tools/probe_l2.hip).  Placement affects speed only.
IMT_DEVICE int imt_xcd_block(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---------------------------------------------------------------- scalar conversions
template <typename T> IMT_DEVICE float to_f32(T v);
template <> IMT_DEVICE float to_f32<float>(float v) { return v; }
template <> IMT_DEVICE float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <typename T> IMT_DEVICE T from_f32(float v);
template <> IMT_DEVICE float from_f32<float>(float v) { return v; }
template <> IMT_DEVICE bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// 4-element vector of T <-> f32x4 (global/LDS, 8- or 16-byte accesses)
#ifndef IMT_WT_STORES
#define IMT_WT_STORES 0
#endif
template <typename T> struct Vec4;
template <> struct Vec4<float> {
  typedef f32x4 type;
  static IMT_DEVICE f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static IMT_DEVICE void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
  // write-through (sc1) store: the bytes leave the XCD's L2 as the store completes -- for data that another workgroup of the
  // SAME launch reads (imt_gemm's in-launch LayerNorm); two 8-byte stores (the parity mode, not the fast path)
  static IMT_DEVICE void store_wt(float* p, f32x4 v) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, __builtin_bit_cast(unsigned long long, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  static IMT_DEVICE type load_raw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }  // no conversion: the
  static IMT_DEVICE f32x4 cvt(type v) { return v; }                                                // load stays in flight
};
template <> struct Vec4<bf16_t> {
  typedef bf16x4 type;
  static IMT_DEVICE f32x4 load(const bf16_t* p) {
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
  }
  static IMT_DEVICE void store(bf16_t* p, f32x4 v) {
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
#if IMT_WT_STORES  // experiment: write-through (sc1) output stores, nothing left dirty for the end-of-kernel write-back
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, r), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
#else
    *reinterpret_cast<bf16x4*>(p) = r;
#endif
  }
  static IMT_DEVICE void store_wt(bf16_t* p, f32x4 v) {  // global_store_dwordx2 ... sc1
    bf16x4 r = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, r), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  static IMT_DEVICE type load_raw(const bf16_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
  static IMT_DEVICE f32x4 cvt(type v) {
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
  }
};

// ---------------------------------------------------------------- wave / block reductions (wave = 64)
IMT_DEVICE float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
IMT_DEVICE float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------- math
// exact-erf GELU (src/lm_config.py:7 "gelu" == F.gelu) evaluated with the Abramowitz-Stegun 7.1.26 rational form:
//   erf(x) = 1 - (a1 t + ... + a5 t^5) exp(-x^2),  t = 1/(1 + p x),  x >= 0      |error| <= 1.5e-7 (absolute)
// i.e. below fp32 round-off of the surrounding sums and 3 orders under the 1e-4 parity bar, at ~1/4 of the VALU cost
// of libm's erff (the GELU / GELU' epilogues of the FFN GEMMs were costing as much as their MFMA main loops).
// The same exponential serves the Gaussian pdf of the derivative: exp(-x^2) with x = z/sqrt(2) is exp(-z^2/2).
IMT_DEVICE void erf_and_gauss(float z, float& erf_v, float& gauss) {
  const float x = fabsf(z) * 0.70710678118654752440f;
  const float e = __expf(-x * x);
  // v_rcp_f32 (1 ulp) -- __frcp_rn / a plain division expand to the 11-instruction IEEE sequence (div_scale, rcp, five FMAs,
  // div_fmas, div_fixup), 40 % of this function, in epilogues that are VALU-bound (12 us of a 28-us FFN-up GEMM)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float y = 1.0f - p * t * e;
  erf_v = copysignf(y, z);
  gauss = e;
}
// Four elements at a time on the packed fp32 pipe (v_pk_mul / v_pk_fma_f32: two lanes' worth per issue slot): the GELU and
// GELU' epilogues of the FFN GEMMs run no MFMA beside this math and are bound by VALU issue (64 K elements per 256 x 256
// tile); ~13 issue slots per element against ~19 for the scalar form.  Same operations in the same order as the scalar
// functions below (which serve ragged tails).
typedef float f32x2 __attribute__((ext_vector_type(2)));
IMT_DEVICE f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
IMT_DEVICE void erf_and_gauss2(f32x2 z, f32x2& erf_v, f32x2& gauss) {
  const f32x2 x = __builtin_elementwise_abs(z) * 0.70710678118654752440f;
  const f32x2 a = x * x * -1.4426950408889634f;  // exp(-x^2) = exp2(-x^2 log2 e)
  f32x2 e, t;
  e.x = __builtin_amdgcn_exp2f(a.x); e.y = __builtin_amdgcn_exp2f(a.y);
  const f32x2 d = pk_fma(x, f32x2{0.3275911f, 0.3275911f}, f32x2{1.f, 1.f});
  t.x = __builtin_amdgcn_rcpf(d.x); t.y = __builtin_amdgcn_rcpf(d.y);
  // -(a1 t + ... + a5 t^5) / t with the signs folded into the coefficients
  f32x2 p = pk_fma(t, f32x2{-1.061405429f, -1.061405429f}, f32x2{1.453152027f, 1.453152027f});
  p = pk_fma(p, t, f32x2{-1.421413741f, -1.421413741f});
  p = pk_fma(p, t, f32x2{0.284496736f, 0.284496736f});
  p = pk_fma(p, t, f32x2{-0.254829592f, -0.254829592f});
  const f32x2 y = pk_fma(p * t, e, f32x2{1.f, 1.f});
  erf_v.x = copysignf(y.x, z.x); erf_v.y = copysignf(y.y, z.y);
  gauss = e;
}
IMT_DEVICE f32x4 gelu_erf4(f32x4 v) {
  f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]}, e0, g0, e1, g1;
  erf_and_gauss2(lo, e0, g0);
  erf_and_gauss2(hi, e1, g1);
  lo = lo * 0.5f * (e0 + 1.0f);
  hi = hi * 0.5f * (e1 + 1.0f);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
IMT_DEVICE f32x4 gelu_erf_grad4(f32x4 z) {
  f32x2 lo = {z[0], z[1]}, hi = {z[2], z[3]}, e0, g0, e1, g1;
  erf_and_gauss2(lo, e0, g0);
  erf_and_gauss2(hi, e1, g1);
  lo = pk_fma(lo * 0.39894228040143267794f, g0, (e0 + 1.0f) * 0.5f);
  hi = pk_fma(hi * 0.39894228040143267794f, g1, (e1 + 1.0f) * 0.5f);
  return f32x4{lo.x, lo.y, hi.x, hi.y};
}
IMT_DEVICE float gelu_erf(float z) {
  float er, ga;
  erf_and_gauss(z, er, ga);
  return 0.5f * z * (1.0f + er);
}
IMT_DEVICE float gelu_erf_grad(float z) {
  float er, ga;
  erf_and_gauss(z, er, ga);
  return 0.5f * (1.0f + er) + z * 0.39894228040143267794f * ga;
}

// ---------------------------------------------------------------- counter-based dropout RNG
// Deterministic Bernoulli(1-p) per element, recomputed identically in backward.  One call of the 32-bit finaliser below
// (lowbias32: two multiplies, three xor-shifts -- a complete avalanche by itself) yields 32 good bits = TWO 16-bit
// draws; the decision of element idx of a row-major tensor is draw (idx & 3) of the block (idx >> 2): two finaliser
// calls per FOUR consecutive elements, which is what every lane of the GEMM / LayerNorm / embedding kernels holds
// (round 1 spent two calls -- four quarter-rate multiplies -- per element; the attention and epilogue element phases
// are VALU-bound).  p is realised as round(p * 65536) / 65536 (0.1 -> 0.100006).
IMT_DEVICE uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
IMT_DEVICE uint32_t dropout_key(uint64_t seed) { return (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9e3779b9U); }
// the 4 x 16 draws of block idx4 (= element index >> 2) as two words
IMT_DEVICE void dropout_block(uint32_t key, uint64_t idx4, uint32_t& h_lo, uint32_t& h_hi) {
  h_lo = mix32(((uint32_t)idx4 ^ key) + (uint32_t)(idx4 >> 32) * 0x85ebca6bU);
  h_hi = mix32(h_lo + 0x9e3779b9U);
}
IMT_DEVICE bool dropout_draw(uint32_t h_lo, uint32_t h_hi, int e /* 0..3 */, uint32_t thresh16) {
  const uint32_t w = (e & 2) ? h_hi : h_lo;
  return ((e & 1) ? (w >> 16) : (w & 0xffffu)) >= thresh16;
}
// single element (edge tiles, tails); same decision as the block form
IMT_DEVICE bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh16) {
  uint32_t lo, hi;
  dropout_block(dropout_key(seed), idx >> 2, lo, hi);
  return dropout_draw(lo, hi, (int)(idx & 3), thresh16);
}
// v[e] <- keep(idx0 + e) ? v[e] * inv_keep : 0 for the 4 consecutive elements idx0 .. idx0 + 3, idx0 % 4 == 0
IMT_DEVICE void dropout_apply4(f32x4& v, uint64_t seed, uint64_t idx0, uint32_t thresh16, float inv_keep) {
  uint32_t lo, hi;
  dropout_block(dropout_key(seed), idx0 >> 2, lo, hi);
  v[0] = (lo & 0xffffu) >= thresh16 ? v[0] * inv_keep : 0.f;
  v[1] = (lo >> 16) >= thresh16 ? v[1] * inv_keep : 0.f;
  v[2] = (hi & 0xffffu) >= thresh16 ? v[2] * inv_keep : 0.f;
  v[3] = (hi >> 16) >= thresh16 ? v[3] * inv_keep : 0.f;
}
// attention probabilities: pairs along the key axis share one finaliser call (the forward's lane holds 4 consecutive keys
// of one query: 2 calls per 4 elements; the backward's lane holds 4 consecutive queries of one key: 1 call per element).
// pair index = (row of the [B*H*Tq] x ceil(Tk/2) pair matrix); decision = 16-bit half (j & 1).
IMT_DEVICE uint32_t attn_drop_word(uint32_t key, uint64_t pair_idx) {
  return mix32(((uint32_t)pair_idx ^ key) + (uint32_t)(pair_idx >> 32) * 0x85ebca6bU);
}
IMT_DEVICE uint32_t attn_drop_word(uint32_t key, uint32_t pair_idx) { return mix32(pair_idx ^ key); }  // same word for indices < 2^32
IMT_DEVICE bool attn_drop_keep(uint32_t word, int j, uint32_t thresh16) {
  return ((j & 1) ? (word >> 16) : (word & 0xffffu)) >= thresh16;
}
// p -> 16-bit threshold (0 = dropout off)
static inline uint32_t dropout_thresh(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 65536.0 + 0.5;
  if (t > 65535.0) t = 65535.0;
  if (t < 1.0) t = 1.0;
  return (uint32_t)t;
}
