// Single-query attention over a slot-addressed K/V cache: one WAVE per (hypothesis, head).  Shared by the per-operator launch
// (decode.hip: attn_decode_kernel) and the one-launch decoder step (decode_fused.hip).
#pragma once
#include "common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------ attention
// One wave per (hypothesis, head).  A key/value row of the head is DH elements = DH/8 lanes x 8 elements (16 B for
// bf16), so a wave streams 64/(DH/8) keys per iteration with fully coalesced row segments.  Each lane group keeps
// its own running (max, sum, o[8]) -- online softmax -- and the groups are merged once at the end.
template <typename T> IMT_DEVICE void load8(const T* p, float (&v)[8]);
template <> IMT_DEVICE void load8<float>(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <> IMT_DEVICE void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
}
template <typename T> IMT_DEVICE void store8(T* p, const float (&v)[8]);
template <> IMT_DEVICE void store8<float>(float* p, const float (&v)[8]) {
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <> IMT_DEVICE void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  bf16x8 a;
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = (bf16_t)v[e];
  *reinterpret_cast<bf16x8*>(p) = a;
}

// w: index of the (hypothesis, head) pair, w < a.R * a.H
// eight consecutive elements of a row as they sit in memory, converted when used
template <typename T> struct Raw8;
template <> struct Raw8<float> {
  struct type { f32x4 a, b; };
  static IMT_DEVICE type load(const float* p) { return type{*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4)}; }
  static IMT_DEVICE void cvt(const type& r, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = r.a[e]; v[4 + e] = r.b[e]; }
  }
};
template <> struct Raw8<bf16_t> {
  typedef bf16x8 type;
  static IMT_DEVICE type load(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static IMT_DEVICE void cvt(const type& r, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)r[e];
  }
};

// WT: the output is read by another workgroup of the SAME launch (decode_fused.hip): write-through stores
template <typename T, int DH, bool WT = false, int UNR_ = 0>
IMT_DEVICE void attn_decode_wave(const imt_attn_decode_args& a, int w) {
  constexpr int CH = DH / 8;    // lanes per key row
  constexpr int G = 64 / CH;    // keys in flight per wave
  const int lane = threadIdx.x & 63;
  const int r = w / a.H, h = w % a.H;
  const int c = lane % CH, g = lane / CH;
  const int sent = r / a.rep;
  float q[8];
  load8<T>(reinterpret_cast<const T*>(a.Q) + (int64_t)r * a.ldq + h * DH + 8 * c, q);
  const T* Kb = reinterpret_cast<const T*>(a.K) + h * DH + 8 * c;
  const T* Vb = reinterpret_cast<const T*>(a.V) + h * DH + 8 * c;
  float m = -INFINITY, l = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // UNR key groups per iteration: their slot lookups, then their K/V row segments, are all in flight before the first
  // dependent softmax update (one group per iteration was a chain of exposed load latencies: 18.7 us at 145 keys)
  // bf16 rows wait in registers as loaded (16 bytes = 4 registers each), so more groups fit: six = 48 keys of a 64-wide head per
  // iteration at 124 registers (four waves per SIMD; eight groups take 134 and drop to three).  (The groups are consumed in the same
  // order whatever UNR is: results do not depend on it.)
  constexpr int UNR = UNR_ ? UNR_ : (sizeof(T) == 2 ? 6 : 4);   // (UNR_: the one-launch step runs 16 waves per CU on 128 registers)
  for (int j0 = 0; j0 < a.n_keys; j0 += G * UNR) {
    typename Raw8<T>::type k[UNR], v[UNR];
    int jj[UNR];
    int64_t off[UNR];
    uint8_t mk[UNR];
    // slot and mask lookups are unconditional loads (an absent table reads a valid stand-in address and the value is
    // replaced by a select): a load under a branch drains vmcnt at the join, and the mask byte used to be fetched
    // inside the dependent softmax chain below -- one exposed latency per key group
    const int32_t* slot_p = a.slots ? a.slots + (int64_t)r * a.ld_slots : reinterpret_cast<const int32_t*>(a.Q);
    const uint8_t* mask_p = a.key_mask ? a.key_mask + (int64_t)sent * a.ld_mask : reinterpret_cast<const uint8_t*>(a.Q);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * G + g;
      jj[u] = j < a.n_keys ? j : a.n_keys - 1;
      const int32_t sv = slot_p[a.slots ? jj[u] : 0];
      mk[u] = mask_p[a.key_mask ? jj[u] : 0];
      const int64_t row = a.slots ? (int64_t)sv : (int64_t)sent;
      off[u] = row * a.ld_row + (int64_t)jj[u] * a.ld_pos;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      k[u] = Raw8<T>::load(Kb + off[u]);
      v[u] = Raw8<T>::load(Vb + off[u]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool valid = j0 + u * G + g < a.n_keys;
      float s = 0.f, kf[8], vf[8];
      Raw8<T>::cvt(k[u], kf);
      Raw8<T>::cvt(v[u], vf);
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(q[e], kf[e], s);
#pragma unroll
      for (int x = 1; x < CH; x <<= 1) s += __shfl_xor(s, x, 64);
      s *= a.scale;
      if (a.key_mask && !mk[u]) s += -10000.0f;
      if (valid) {
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), p = __expf(s - mn);  // exp(-inf) == 0 on the first key
        l = l * corr + p;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(p, vf[e], o[e] * corr);
        m = mn;
      }
    }
  }
  // merge the G lane groups (lanes with equal c)
  float M = m;
#pragma unroll
  for (int x = CH; x < 64; x <<= 1) M = fmaxf(M, __shfl_xor(M, x, 64));
  const float f = (m == -INFINITY) ? 0.f : __expf(m - M);
  l *= f;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] *= f;
#pragma unroll
  for (int x = CH; x < 64; x <<= 1) {
    l += __shfl_xor(l, x, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += __shfl_xor(o[e], x, 64);
  }
  if (g == 0) {
    const float inv = 1.0f / l;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= inv;
    T* op = reinterpret_cast<T*>(a.O) + (int64_t)r * a.ldo + h * DH + 8 * c;
    if (WT && sizeof(T) == 2) {
      Vec4<T>::store_wt(op, f32x4{o[0], o[1], o[2], o[3]});
      Vec4<T>::store_wt(op + 4, f32x4{o[4], o[5], o[6], o[7]});
    } else {
      store8<T>(op, o);
    }
  }
}

template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_decode_kernel(imt_attn_decode_args a) {
  const int w = imt_xcd_block(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (w >= a.R * a.H) return;
  attn_decode_wave<T, DH>(a, w);
}

}  // namespace
