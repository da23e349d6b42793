// Fused multi-head attention forward / backward (HF BertSelfAttention semantics):
//   P = softmax(Q K^T * scale + (1 - mask) * -10000) ; [dropout] ; O = P V
// Flash-style: the [Tq,Tk] score matrix never reaches HBM; forward saves one log-sum-exp per query row,
// backward recomputes P from it.
//
// Layout of the work (wave64, 16x16 MFMA tiles, see mma.hpp):
//   fwd / dQ  : one workgroup = 64 query rows of one (batch, head); each of its 4 waves owns 16 queries with
//               the QUERY on the MFMA lane (S^T = K Q^T), so row max / row sum / rescale are lane-local plus
//               two cross-lane shuffles, and the S^T accumulators are directly the B operand of
//               O^T = V^T P^T (resp. dQ^T = K^T dS^T).  K/V tiles of 64 keys are staged in swizzled LDS;
//               V (resp. K) is consumed through the transposed LDS read.
//   dK/dV     : one workgroup = 64 keys; KEY on the lane (S = Q K^T), dV^T / dK^T accumulate in registers over
//               all query tiles; Q and dO tiles are staged once and read both row-wise and transposed.
// Nothing is summed across workgroups: no atomics, bitwise reproducible.
#include <stdlib.h>
#include "mma.hpp"

namespace {

struct AttnP {
  int B, H, Tq, Tk;
  const void* Q; int64_t ldq;
  const void* K; int64_t ldk;
  const void* V; int64_t ldv;
  void* O; int64_t ldo;
  float* lse;
  const uint8_t* key_mask;
  const uint8_t* query_mask;
  const uint8_t* mask3d;
  int causal;
  float scale;
  uint32_t drop_thresh; float inv_keep; uint64_t seed;
  const void* dO; int64_t lddo;
  void* dQ; int64_t lddq;
  void* dK; int64_t lddk;
  void* dV; int64_t lddv;
  unsigned long long* trace;  // tuning only (IMT_TRACE)
  float* delta;
};

// stage a [64 rows][DH] tile (rows row0.. of one (b,h)) into swizzled LDS, zero-filling rows >= nrows
template <typename T, int DH>
IMT_DEVICE void stage_tile(char* tile, const T* base, int64_t ld, int row0, int nrows) {
  constexpr int RB = DH * sizeof(T), CPR = RB / 16, EPC = 16 / sizeof(T);
  // (Kept as a loop with the load under its bounds test: hoisting all loads of a tile into registers, as the short-sequence
  // kernels do in their prologues, costs the register-bound dK/dV kernel 200 spilled registers -- C4 2.1 -> 2.8 ms.)
  for (int q = threadIdx.x; q < 64 * CPR; q += 256) {
    const int tr = q / CPR, c = q % CPR;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row0 + tr < nrows) v = *reinterpret_cast<const u32x4*>(base + (int64_t)(row0 + tr) * ld + c * EPC);
    *reinterpret_cast<u32x4*>(tile + tile_off<RB>(tr, c)) = v;
  }
}

// load NS register fragments of row (row0 + lane&15) straight from global (K-contiguous operand)
template <typename T, int DH>
IMT_DEVICE void load_row_frags(typename Frag<T>::type (&f)[DH * sizeof(T) / 64], const T* base, int64_t ld, int row0,
                               int nrows) {
  constexpr int NS = DH * sizeof(T) / 64, EPC = 16 / sizeof(T);
  const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
  typedef typename Frag<T>::type frag_t;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    const u32x4 v = *reinterpret_cast<const u32x4*>(base + (int64_t)min(row0 + r, nrows - 1) * ld + (4 * s + g) * EPC);
    f[s] = __builtin_bit_cast(frag_t, (row0 + r < nrows) ? v : z);  // unconditional load + select (see stage_tile)
  }
}

// MASK3D = false: the kernel instance for launches without an explicit [B, Tq, Tk] mask -- the per-element test of the
// pointer is a branch + exec save/restore + wait skeleton around a global load, ~10 scalar instructions per element
// even when the pointer is null.
template <bool MASK3D = true>
IMT_DEVICE bool mask_ok(const AttnP& p, int b, int i, int j, bool key_ok, bool query_ok) {
  bool ok = key_ok && query_ok;
  if (p.causal) ok = ok && (j <= i);
  if (MASK3D) {
    if (p.mask3d && i < p.Tq && j < p.Tk) ok = ok && (p.mask3d[((int64_t)b * p.Tq + i) * p.Tk + j] != 0);
  }
  return ok;
}
// Dropout of the attention probabilities: attn_drop_word / attn_drop_keep (common.hpp) -- one finaliser call per PAIR of
// neighbouring keys of a query; pair index = ((b*H + h)*Tq + i) * ceil(Tk/2) + (j >> 1).
IMT_DEVICE uint64_t attn_pair_row(const AttnP& p, int b, int h, int i) {
  return ((uint64_t)(b * p.H + h) * (uint64_t)p.Tq + (uint64_t)i) * (uint64_t)((p.Tk + 1) >> 1);
}

// accumulator tiles -> operand fragments of the following product (see mma.hpp, "permuted-K")
template <typename T> struct AccOperand;
template <> struct AccOperand<float> {
  static constexpr int NFR = 4;  // one fragment per 16-row tile
  static IMT_DEVICE f32x4 get(const f32x4 (&t)[4], int u) { return t[u]; }
  template <int RB> static IMT_DEVICE f32x4 lds(const char* tile, int u, int col0) { return lds_frag_kperm_f32<RB>(tile, 16 * u, col0); }
};
template <> struct AccOperand<bf16_t> {
  static constexpr int NFR = 2;  // one fragment per PAIR of 16-row tiles
  static IMT_DEVICE bf16x8 get(const f32x4 (&t)[4], int u) { return acc_pair_to_frag(t[2 * u], t[2 * u + 1]); }
  template <int RB> static IMT_DEVICE bf16x8 lds(const char* tile, int u, int col0) { return lds_frag_kperm_bf16<RB>(tile, 32 * u, col0); }
};

// =========================================================================================== forward
template <typename T, int DH, bool MASK3D>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 4 : 3) void attn_fwd_kernel(AttnP p) {
  constexpr int RB = DH * sizeof(T), NS = RB / 64, NDT = DH / 16;
  typedef typename Frag<T>::type frag_t;
  constexpr int NSUB = (RB > 128) ? 1 : 2;  // 64-row sub-tiles staged per barrier pair (LDS budget)
  __shared__ __attribute__((aligned(16))) char Ks[NSUB][64 * RB];
  __shared__ __attribute__((aligned(16))) char Vs[NSUB][64 * RB];
  __shared__ uint8_t kmask_s[NSUB * 64];

  const int nqt = (p.Tq + 63) / 64;
  const int lid = imt_xcd_block(blockIdx.x, gridDim.x);  // batch-major: XCD x owns batches [x*B/8, (x+1)*B/8)
  const int b = lid / (p.H * nqt), h = (lid / nqt) % p.H, tile_x = lid % nqt;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int q0 = tile_x * 64 + wave * 16;
  const int i = q0 + r;  // this lane's query row
  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)b * p.Tq * p.ldq + h * DH;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)b * p.Tk * p.ldk + h * DH;
  const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)b * p.Tk * p.ldv + h * DH;

  frag_t qf[NS];
  load_row_frags<T, DH>(qf, Qb, p.ldq, q0, p.Tq);
  const bool query_ok = (p.query_mask && i < p.Tq) ? (p.query_mask[(int64_t)b * p.Tq + i] != 0) : true;

  f32x4 o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int nkt = (p.Tk + 63) / 64;
  for (int kt0 = 0; kt0 < nkt; kt0 += NSUB) {
    __syncthreads();  // previous tiles fully consumed
    // all the loads of up to NSUB key tiles (K, V, mask) are in flight together: one memory round trip per 128 keys
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      if (kt0 + sub < nkt) {
        stage_tile<T, DH>(Ks[sub], Kb, p.ldk, (kt0 + sub) * 64, p.Tk);
        stage_tile<T, DH>(Vs[sub], Vb, p.ldv, (kt0 + sub) * 64, p.Tk);
      }
    }
    if (threadIdx.x < NSUB * 64) {
      const int j = kt0 * 64 + threadIdx.x;
      kmask_s[threadIdx.x] = (j < p.Tk) ? (p.key_mask ? p.key_mask[(int64_t)b * p.Tk + j] : (uint8_t)1) : (uint8_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
    const int kt = kt0 + sub;
    if (kt >= nkt) break;
    const char* Kt = Ks[sub];
    const char* Vt = Vs[sub];
    const uint8_t* kmask_t = kmask_s + 64 * sub;

    // S^T tiles: rows <- keys (16 per tile), cols <- this wave's 16 queries
    f32x4 s[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) mma16(s[nt], lds_frag_kcontig<T, RB>(Kt, 16 * nt, 4 * ks), qf[ks]);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int jl = 16 * nt + 4 * g + e, j = kt * 64 + jl;
        float v = s[nt][e] * p.scale + (mask_ok<MASK3D>(p, b, i, j, kmask_t[jl] != 0, query_ok) ? 0.f : -10000.0f);
        if (j >= p.Tk) v = -INFINITY;
        s[nt][e] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = __expf(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pv = __expf(s[nt][e] - m_new);
        psum += pv;
        s[nt][e] = pv;
      }
    if (p.drop_thresh) {
      const uint32_t key = dropout_key(p.seed);
      const uint64_t prow = attn_pair_row(p, b, h, i) + (uint64_t)((kt * 64 + 4 * g) >> 1);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {  // keys 16nt + 4g + 2e2, +1: one pair
          const uint32_t w = attn_drop_word(key, prow + 8 * nt + e2);
          s[nt][2 * e2] = (w & 0xffffu) >= p.drop_thresh ? s[nt][2 * e2] * p.inv_keep : 0.f;
          s[nt][2 * e2 + 1] = (w >> 16) >= p.drop_thresh ? s[nt][2 * e2 + 1] * p.inv_keep : 0.f;
        }
    }
    l_run = l_run * alpha + psum;  // per-lane partial; lanes r, r+16, r+32, r+48 are combined at the end
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
    // O^T[d][m] += V^T[d][key] P^T[key][m]
#pragma unroll
    for (int u = 0; u < AccOperand<T>::NFR; ++u) {
      const frag_t pf = AccOperand<T>::get(s, u);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) mma16(o[dt], AccOperand<T>::template lds<RB>(Vt, u, 16 * dt), pf);
    }
    }  // sub
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv_l = 1.0f / l_run;
  if (i < p.Tq) {
    T* Ob = reinterpret_cast<T*>(p.O) + ((int64_t)b * p.Tq + i) * p.ldo + h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) Vec4<T>::store(Ob + 16 * dt + 4 * g, o[dt] * inv_l);
    if (g == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.Tq + i] = m_run + __logf(l_run);
  }
}

// =========================================================================================== forward, short sequences
// Tq, Tk <= 128, bf16: one workgroup (8 waves, 16 queries each) per (batch, head).  K and V are staged ONCE for all
// 128 queries (the 64-query kernel above stages them once per query tile), and with every key resident the softmax is
// a single pass: all S^T tiles, one max, one exp/sum, then O^T = V^T P^T -- no running rescale.
template <int DH, bool MASK3D>
__global__ __launch_bounds__(512, 4) void attn_fwd_short_kernel(AttnP p) {
  typedef bf16_t T;
  constexpr int RB = DH * 2, NS = RB / 64, NDT = DH / 16;
  typedef typename Frag<T>::type frag_t;
  __shared__ __attribute__((aligned(16))) char Ks[128 * RB];
  __shared__ __attribute__((aligned(16))) char Vs[128 * RB];
  __shared__ uint8_t kmask_s[128];
  const int lid = imt_xcd_block(blockIdx.x, gridDim.x);
  const int b = lid / p.H, h = lid % p.H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int q0 = wave * 16, i = q0 + r;
  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)b * p.Tq * p.ldq + h * DH;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)b * p.Tk * p.ldk + h * DH;
  const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)b * p.Tk * p.ldv + h * DH;
  // every global load of the kernel is requested here, unconditionally (clamped rows, a valid stand-in address for an
  // absent mask) and before the first use: ONE memory round trip in front of the barrier instead of one per load
  IMT_STAMP(p.trace, 0);
  constexpr int CPR = RB / 16, NIT = 128 * CPR / 512;
  u32x4 vk[NIT], vv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int q = threadIdx.x + it * 512, tr = min(q / CPR, p.Tk - 1), c = q % CPR;
    vk[it] = *reinterpret_cast<const u32x4*>(Kb + (int64_t)tr * p.ldk + c * 8);
    vv[it] = *reinterpret_cast<const u32x4*>(Vb + (int64_t)tr * p.ldv + c * 8);
  }
  frag_t qf[NS];
  load_row_frags<T, DH>(qf, Qb, p.ldq, q0, p.Tq);
  const uint8_t* kmp = p.key_mask ? p.key_mask + (int64_t)b * p.Tk + min((int)threadIdx.x & 127, p.Tk - 1) : reinterpret_cast<const uint8_t*>(Kb);
  const uint8_t* qmp = p.query_mask ? p.query_mask + (int64_t)b * p.Tq + min(i, p.Tq - 1) : reinterpret_cast<const uint8_t*>(Qb);
  const uint8_t kmv = *kmp, qmv = *qmp;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int q = threadIdx.x + it * 512, tr = q / CPR, c = q % CPR;
    const u32x4 z = {0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(Ks + tile_off<RB>(tr, c)) = (tr < p.Tk) ? vk[it] : z;
    *reinterpret_cast<u32x4*>(Vs + tile_off<RB>(tr, c)) = (tr < p.Tk) ? vv[it] : z;
  }
  if (threadIdx.x < 128) {
    const int j = threadIdx.x;
    kmask_s[j] = (j < p.Tk) ? (p.key_mask ? kmv : (uint8_t)1) : (uint8_t)0;
  }
  const bool query_ok = (p.query_mask && i < p.Tq) ? (qmv != 0) : true;
  __syncthreads();
  IMT_STAMP(p.trace, 1);
  if (q0 >= p.Tq) return;  // whole wave past the last query (no barrier follows)

  // S^T: 8 tiles of 16 keys x this wave's 16 queries
  f32x4 s[8];
  float tmax = -INFINITY;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) mma16(s[nt], lds_frag_kcontig<T, RB>(Ks, 16 * nt, 4 * ks), qf[ks]);
    const uint32_t km4 = *reinterpret_cast<const uint32_t*>(kmask_s + 16 * nt + 4 * g);  // this lane's 4 keys
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 16 * nt + 4 * g + e;
      float v = s[nt][e] * p.scale + (mask_ok<MASK3D>(p, b, i, j, ((km4 >> (8 * e)) & 0xffu) != 0, query_ok) ? 0.f : -10000.0f);
      if (j >= p.Tk) v = -INFINITY;
      s[nt][e] = v;
      tmax = fmaxf(tmax, v);
    }
  }
  tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
  IMT_STAMP(p.trace, 2);
  float psum = 0.f;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pv = __expf(s[nt][e] - tmax);
      psum += pv;
      s[nt][e] = pv;
    }
  if (p.drop_thresh) {  // one uniform branch around all 32 elements; 16 finaliser calls for them (pairs of keys)
    const uint32_t key = dropout_key(p.seed);
    const uint32_t prow = (uint32_t)attn_pair_row(p, b, h, i) + (uint32_t)(2 * g);  // (host: B*H*Tq*Tk < 2^32 for this kernel)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const uint32_t w = attn_drop_word(key, prow + 8 * nt + e2);
        s[nt][2 * e2] = (w & 0xffffu) >= p.drop_thresh ? s[nt][2 * e2] * p.inv_keep : 0.f;
        s[nt][2 * e2 + 1] = (w >> 16) >= p.drop_thresh ? s[nt][2 * e2 + 1] * p.inv_keep : 0.f;
      }
  }
  psum += __shfl_xor(psum, 16, 64);
  psum += __shfl_xor(psum, 32, 64);
  IMT_STAMP(p.trace, 3);
  f32x4 o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u) {  // 32 keys per MFMA k-step
    const frag_t pf = acc_pair_to_frag(s[2 * u], s[2 * u + 1]);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) mma16(o[dt], lds_frag_kperm_bf16<RB>(Vs, 32 * u, 16 * dt), pf);
  }
  const float inv_l = 1.0f / psum;
  if (i < p.Tq) {
    T* Ob = reinterpret_cast<T*>(p.O) + ((int64_t)b * p.Tq + i) * p.ldo + h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) Vec4<T>::store(Ob + 16 * dt + 4 * g, o[dt] * inv_l);
    if (g == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.Tq + i] = tmax + __logf(psum);
  }
  IMT_STAMP(p.trace, 4);
}

// =========================================================================================== backward: dQ
template <typename T, int DH, bool MASK3D>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 4 : 3) void attn_bwd_dq_kernel(AttnP p) {
  constexpr int RB = DH * sizeof(T), NS = RB / 64, NDT = DH / 16;
  typedef typename Frag<T>::type frag_t;
  constexpr int NSUB = (RB > 128) ? 1 : 2;
  __shared__ __attribute__((aligned(16))) char Ks[NSUB][64 * RB];
  __shared__ __attribute__((aligned(16))) char Vs[NSUB][64 * RB];
  __shared__ uint8_t kmask_s[NSUB * 64];

  const int nqt = (p.Tq + 63) / 64;
  const int lid = imt_xcd_block(blockIdx.x, gridDim.x);  // batch-major: XCD x owns batches [x*B/8, (x+1)*B/8)
  const int b = lid / (p.H * nqt), h = (lid / nqt) % p.H, tile_x = lid % nqt;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int q0 = tile_x * 64 + wave * 16;
  const int i = q0 + r;
  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)b * p.Tq * p.ldq + h * DH;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)b * p.Tk * p.ldk + h * DH;
  const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)b * p.Tk * p.ldv + h * DH;
  const T* dOb = reinterpret_cast<const T*>(p.dO) + (int64_t)b * p.Tq * p.lddo + h * DH;

  frag_t qf[NS], dof[NS];
  load_row_frags<T, DH>(qf, Qb, p.ldq, q0, p.Tq);
  load_row_frags<T, DH>(dof, dOb, p.lddo, q0, p.Tq);
  const bool query_ok = (p.query_mask && i < p.Tq) ? (p.query_mask[(int64_t)b * p.Tq + i] != 0) : true;
  const int64_t srow = ((int64_t)b * p.H + h) * p.Tq + (i < p.Tq ? i : 0);
  const float lse_i = p.lse[srow];
  // delta_i = rowsum(dO * O) of this lane's query row, computed here from the row fragments (the lanes r, r+16, r+32,
  // r+48 hold the four 16-byte chunks of each 64-byte k-step) and published for the dK/dV kernel that runs next.
  float delta_i = 0.f;
  {
    const T* Ob = reinterpret_cast<const T*>(p.O) + (int64_t)b * p.Tq * p.ldo + h * DH;
    frag_t of[NS];
    load_row_frags<T, DH>(of, Ob, p.ldo, q0, p.Tq);
#pragma unroll
    for (int ks = 0; ks < NS; ++ks)
#pragma unroll
      for (int e = 0; e < Frag<T>::EPC; ++e) delta_i += (float)dof[ks][e] * (float)of[ks][e];
    delta_i += __shfl_xor(delta_i, 16, 64);
    delta_i += __shfl_xor(delta_i, 32, 64);
    if (g == 0 && i < p.Tq) p.delta[srow] = delta_i;
  }

  f32x4 dq[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (p.Tk + 63) / 64;
  for (int kt0 = 0; kt0 < nkt; kt0 += NSUB) {
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      if (kt0 + sub < nkt) {
        stage_tile<T, DH>(Ks[sub], Kb, p.ldk, (kt0 + sub) * 64, p.Tk);
        stage_tile<T, DH>(Vs[sub], Vb, p.ldv, (kt0 + sub) * 64, p.Tk);
      }
    }
    if (threadIdx.x < NSUB * 64) {
      const int j = kt0 * 64 + threadIdx.x;
      kmask_s[threadIdx.x] = (j < p.Tk) ? (p.key_mask ? p.key_mask[(int64_t)b * p.Tk + j] : (uint8_t)1) : (uint8_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
    const int kt = kt0 + sub;
    if (kt >= nkt) break;
    const char* Kt = Ks[sub];
    const char* Vt = Vs[sub];
    const uint8_t* kmask_t = kmask_s + 64 * sub;
    f32x4 s[4], dp[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dp[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) {
        mma16(s[nt], lds_frag_kcontig<T, RB>(Kt, 16 * nt, 4 * ks), qf[ks]);
        mma16(dp[nt], lds_frag_kcontig<T, RB>(Vt, 16 * nt, 4 * ks), dof[ks]);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int jl = 16 * nt + 4 * g + e, j = kt * 64 + jl;
        float v = s[nt][e] * p.scale + (mask_ok<MASK3D>(p, b, i, j, kmask_t[jl] != 0, query_ok) ? 0.f : -10000.0f);
        float pv = (j < p.Tk) ? __expf(v - lse_i) : 0.f;
        float dpv = dp[nt][e];
        if (p.drop_thresh)
          dpv = attn_drop_keep(attn_drop_word(dropout_key(p.seed), attn_pair_row(p, b, h, i) + (uint64_t)(j >> 1)), j, p.drop_thresh) ? dpv * p.inv_keep : 0.f;
        s[nt][e] = pv * (dpv - delta_i);  // dS^T
      }
#pragma unroll
    for (int u = 0; u < AccOperand<T>::NFR; ++u) {
      const frag_t dsf = AccOperand<T>::get(s, u);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) mma16(dq[dt], AccOperand<T>::template lds<RB>(Kt, u, 16 * dt), dsf);
    }
    }  // sub
  }
  if (i < p.Tq) {
    T* dQb = reinterpret_cast<T*>(p.dQ) + ((int64_t)b * p.Tq + i) * p.lddq + h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) Vec4<T>::store(dQb + 16 * dt + 4 * g, dq[dt] * p.scale);
  }
}

// =========================================================================================== backward: dK, dV
template <typename T, int DH, bool MASK3D>
__global__ __launch_bounds__(256, (DH == 64 && sizeof(T) == 2) ? 3 : 2) void attn_bwd_dkdv_kernel(AttnP p) {
  constexpr int RB = DH * sizeof(T), NS = RB / 64, NDT = DH / 16;
  typedef typename Frag<T>::type frag_t;
  constexpr int NSUB = (RB > 128) ? 1 : 2;
  __shared__ __attribute__((aligned(16))) char Qs[NSUB][64 * RB];
  __shared__ __attribute__((aligned(16))) char dOs[NSUB][64 * RB];
  __shared__ float lse_s[NSUB * 64], delta_s[NSUB * 64];
  __shared__ uint8_t qmask_s[NSUB * 64];

  const int nkt0 = (p.Tk + 63) / 64;
  const int lid = imt_xcd_block(blockIdx.x, gridDim.x);
  const int b = lid / (p.H * nkt0), h = (lid / nkt0) % p.H, tile_x = lid % nkt0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int k0 = tile_x * 64 + wave * 16;
  const int j = k0 + r;  // this lane's key
  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)b * p.Tq * p.ldq + h * DH;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)b * p.Tk * p.ldk + h * DH;
  const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)b * p.Tk * p.ldv + h * DH;
  const T* dOb = reinterpret_cast<const T*>(p.dO) + (int64_t)b * p.Tq * p.lddo + h * DH;

  frag_t kf[NS], vf[NS];
  load_row_frags<T, DH>(kf, Kb, p.ldk, k0, p.Tk);
  load_row_frags<T, DH>(vf, Vb, p.ldv, k0, p.Tk);
  const bool key_ok = (j < p.Tk) ? (p.key_mask ? (p.key_mask[(int64_t)b * p.Tk + j] != 0) : true) : false;

  f32x4 dk[NDT], dv[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nqt = (p.Tq + 63) / 64;
  for (int qt0 = 0; qt0 < nqt; qt0 += NSUB) {
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      if (qt0 + sub < nqt) {
        stage_tile<T, DH>(Qs[sub], Qb, p.ldq, (qt0 + sub) * 64, p.Tq);
        stage_tile<T, DH>(dOs[sub], dOb, p.lddo, (qt0 + sub) * 64, p.Tq);
      }
    }
    if (threadIdx.x < NSUB * 64) {
      const int i = qt0 * 64 + threadIdx.x;
      const int64_t srow = ((int64_t)b * p.H + h) * p.Tq + (i < p.Tq ? i : 0);
      lse_s[threadIdx.x] = p.lse[srow];
      delta_s[threadIdx.x] = p.delta[srow];
      qmask_s[threadIdx.x] = (i < p.Tq) ? (p.query_mask ? p.query_mask[(int64_t)b * p.Tq + i] : (uint8_t)1) : (uint8_t)1;
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
    const int qt = qt0 + sub;
    if (qt >= nqt) break;
    const char* Qt = Qs[sub];
    const char* dOt = dOs[sub];
    const float* lse_t = lse_s + 64 * sub;
    const float* delta_t = delta_s + 64 * sub;
    const uint8_t* qmask_t = qmask_s + 64 * sub;
    // S tiles: rows <- queries (16 per tile), cols <- this wave's 16 keys
    f32x4 s[4], dp[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      s[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dp[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) {
        mma16(s[mt], lds_frag_kcontig<T, RB>(Qt, 16 * mt, 4 * ks), kf[ks]);
        mma16(dp[mt], lds_frag_kcontig<T, RB>(dOt, 16 * mt, 4 * ks), vf[ks]);
      }
    }
    f32x4 pd[4];  // dropped P (operand of dV)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int il = 16 * mt + 4 * g + e, i = qt * 64 + il;
        float v = s[mt][e] * p.scale + (mask_ok<MASK3D>(p, b, i, j, key_ok, qmask_t[il] != 0) ? 0.f : -10000.0f);
        float pv = (i < p.Tq && j < p.Tk) ? __expf(v - lse_t[il]) : 0.f;
        float dpv = dp[mt][e], pdv = pv;
        if (p.drop_thresh) {
          const bool keep = attn_drop_keep(attn_drop_word(dropout_key(p.seed), attn_pair_row(p, b, h, i) + (uint64_t)(j >> 1)), j, p.drop_thresh);
          dpv = keep ? dpv * p.inv_keep : 0.f;
          pdv = keep ? pv * p.inv_keep : 0.f;
        }
        pd[mt][e] = pdv;
        s[mt][e] = pv * (dpv - delta_t[il]);  // dS
      }
#pragma unroll
    for (int u = 0; u < AccOperand<T>::NFR; ++u) {
      const frag_t pf = AccOperand<T>::get(pd, u);
      const frag_t dsf = AccOperand<T>::get(s, u);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        mma16(dv[dt], AccOperand<T>::template lds<RB>(dOt, u, 16 * dt), pf);   // dV^T[d][n] += dO^T[d][m] P[m][n]
        mma16(dk[dt], AccOperand<T>::template lds<RB>(Qt, u, 16 * dt), dsf);   // dK^T[d][n] += Q^T[d][m] dS[m][n]
      }
    }
    }  // sub
  }
  if (j < p.Tk) {
    T* dKb = reinterpret_cast<T*>(p.dK) + ((int64_t)b * p.Tk + j) * p.lddk + h * DH;
    T* dVb = reinterpret_cast<T*>(p.dV) + ((int64_t)b * p.Tk + j) * p.lddv + h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      Vec4<T>::store(dKb + 16 * dt + 4 * g, dk[dt] * p.scale);
      Vec4<T>::store(dVb + 16 * dt + 4 * g, dv[dt]);
    }
  }
}

// =========================================================================================== backward, short sequences
// Tq, Tk <= 128 (the C1 workload: S = T = 128), bf16: ONE workgroup (8 waves) per (batch, head) computes dQ, dK and
// dV together.  The two-kernel path above recomputes S and dP twice, reads Q/K/V/dO/O twice and pays two dependent
// load->compute->store latency chains per attention block; here
//   phase 0: Q and dO (all <= 128 rows) are staged once, delta = rowsum(dO * O) is taken from the row fragments;
//   phase 1: KEY on the lane (wave w owns keys 16w..16w+15, K/V rows in registers): S, P, dP, dS for all queries;
//            dV^T += dO^T P and dK^T += Q^T dS accumulate in registers; dS^T goes to LDS as bf16 [key][query];
//   phase 2: QUERY on the lane (wave w owns queries 16w..): dQ = dS K as a plain K-strided x K-strided product from
//            LDS (dS^T tile, and K parked into the region Q occupied).
// Nothing is summed across workgroups; same dropout mask / mask semantics as the kernels above.
template <int DH, bool MASK3D>
__global__ __launch_bounds__(512, 4) void attn_bwd_fused_kernel(AttnP p) {
  typedef bf16_t T;
  constexpr int RB = DH * 2, NS = RB / 64, NDT = DH / 16;
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;                        // [128][RB]  (phase 2: K)
  char* dOs = Qs + 128 * RB;              // [128][RB]
  char* dST = dOs + 128 * RB;             // [128 keys][128 queries] bf16, 256-B rows
  float* lse_s = reinterpret_cast<float*>(dST + 128 * 256);
  float* delta_s = lse_s + 128;
  uint8_t* qmask_s = reinterpret_cast<uint8_t*>(delta_s + 128);

  const int lid = imt_xcd_block(blockIdx.x, gridDim.x);
  const int b = lid / p.H, h = lid % p.H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)b * p.Tq * p.ldq + h * DH;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)b * p.Tk * p.ldk + h * DH;
  const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)b * p.Tk * p.ldv + h * DH;
  const T* dOb = reinterpret_cast<const T*>(p.dO) + (int64_t)b * p.Tq * p.lddo + h * DH;
  const T* Ob = reinterpret_cast<const T*>(p.O) + (int64_t)b * p.Tq * p.ldo + h * DH;

  // ---- phase 0
  // every global load of the kernel is requested here, unconditionally (clamped rows, a valid stand-in address for an
  // absent mask) and before the first use: one memory round trip in front of the barrier instead of one per load
  IMT_STAMP(p.trace, 0);
  const int k0 = 16 * wave;
  const int j = k0 + r;  // this lane's key (phase 1)
  frag_t kf[NS], vf[NS];
  bool key_ok;
  {
    constexpr int CPR = RB / 16, NIT = 128 * CPR / 512;
    u32x4 vq[NIT], vo[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = threadIdx.x + it * 512, tr = min(q / CPR, p.Tq - 1), c = q % CPR;
      vq[it] = *reinterpret_cast<const u32x4*>(Qb + (int64_t)tr * p.ldq + c * 8);
      vo[it] = *reinterpret_cast<const u32x4*>(dOb + (int64_t)tr * p.lddo + c * 8);
    }
    const int i = 16 * wave + r;  // delta / lse of this wave's 16 query rows
    frag_t of[NS], dof[NS];
    load_row_frags<T, DH>(of, Ob, p.ldo, 16 * wave, p.Tq);
    load_row_frags<T, DH>(dof, dOb, p.lddo, 16 * wave, p.Tq);
    load_row_frags<T, DH>(kf, Kb, p.ldk, k0, p.Tk);
    load_row_frags<T, DH>(vf, Vb, p.ldv, k0, p.Tk);
    const float lse_v = p.lse[((int64_t)b * p.H + h) * p.Tq + min(i, p.Tq - 1)];
    const uint8_t* qmp = p.query_mask ? p.query_mask + (int64_t)b * p.Tq + min(i, p.Tq - 1) : reinterpret_cast<const uint8_t*>(Qb);
    const uint8_t* kmp = p.key_mask ? p.key_mask + (int64_t)b * p.Tk + min(j, p.Tk - 1) : reinterpret_cast<const uint8_t*>(Kb);
    const uint8_t qmv = *qmp, kmv = *kmp;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = threadIdx.x + it * 512, tr = q / CPR, c = q % CPR;
      const u32x4 z = {0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(Qs + tile_off<RB>(tr, c)) = (tr < p.Tq) ? vq[it] : z;
      *reinterpret_cast<u32x4*>(dOs + tile_off<RB>(tr, c)) = (tr < p.Tq) ? vo[it] : z;
    }
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) d += (float)dof[ks][e] * (float)of[ks][e];
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    if (g == 0) {
      delta_s[i] = d;
      lse_s[i] = lse_v;
      qmask_s[i] = (i < p.Tq && p.query_mask) ? qmv : (uint8_t)1;
    }
    key_ok = (j < p.Tk) ? (p.key_mask ? (kmv != 0) : true) : false;
  }
  __syncthreads();
  IMT_STAMP(p.trace, 1);

  // ---- phase 1
  f32x4 dk[NDT], dv[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int nqt = (p.Tq + 63) / 64;
#pragma unroll 1
  for (int qt = 0; qt < 2; ++qt) {
    const char* Qt = Qs + qt * 64 * RB;
    const char* dOt = dOs + qt * 64 * RB;
    f32x4 s[4], dp[4], pd[4];
    if (qt < nqt) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        s[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
          mma16(s[mt], lds_frag_kcontig<T, RB>(Qt, 16 * mt, 4 * ks), kf[ks]);
          mma16(dp[mt], lds_frag_kcontig<T, RB>(dOt, 16 * mt, 4 * ks), vf[ks]);
        }
      }
      // pair index of (query i, this lane's key j) = pair_base + i * ceil(Tk / 2): one finaliser call per element here (the
      // lane's four elements are four different queries)
      const uint32_t dkey = dropout_key(p.seed);
      const uint32_t tkp = (uint32_t)((p.Tk + 1) >> 1);
      const uint32_t pair_base = (uint32_t)(b * p.H + h) * (uint32_t)p.Tq * tkp + (uint32_t)(j >> 1);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        // this lane's 4 consecutive queries: their lse / delta / mask bytes in one LDS read each
        const int i0 = qt * 64 + 16 * mt + 4 * g;
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + i0), del4 = *reinterpret_cast<const f32x4*>(delta_s + i0);
        const uint32_t qm4 = *reinterpret_cast<const uint32_t*>(qmask_s + i0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = i0 + e;
          float v = s[mt][e] * p.scale + (mask_ok<MASK3D>(p, b, i, j, key_ok, ((qm4 >> (8 * e)) & 0xffu) != 0) ? 0.f : -10000.0f);
          float pv = (i < p.Tq && j < p.Tk) ? __expf(v - lse4[e]) : 0.f;
          float dpv = dp[mt][e], pdv = pv;
          if (p.drop_thresh) {  // (host guarantees B*H*Tq*Tk < 2^32 for this kernel)
            const bool keep = attn_drop_keep(attn_drop_word(dkey, pair_base + (uint32_t)i * tkp), j, p.drop_thresh);
            dpv = keep ? dpv * p.inv_keep : 0.f;
            pdv = keep ? pv * p.inv_keep : 0.f;
          }
          pd[mt][e] = pdv;
          s[mt][e] = pv * (dpv - del4[e]);  // dS[i][j]
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const frag_t pf = acc_pair_to_frag(pd[2 * u], pd[2 * u + 1]);
        const frag_t dsf = acc_pair_to_frag(s[2 * u], s[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          mma16(dv[dt], lds_frag_kperm_bf16<RB>(dOt, 32 * u, 16 * dt), pf);   // dV^T[d][key] += dO^T[d][q] P[q][key]
          mma16(dk[dt], lds_frag_kperm_bf16<RB>(Qt, 32 * u, 16 * dt), dsf);   // dK^T[d][key] += Q^T[d][q] dS[q][key]
        }
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) s[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // dS^T[key j][queries 64qt + 16mt + 4g .. +3] (bf16, 8 bytes)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int q_first = qt * 64 + 16 * mt + 4 * g;
      bf16x4 w = {(bf16_t)s[mt][0], (bf16_t)s[mt][1], (bf16_t)s[mt][2], (bf16_t)s[mt][3]};
      *reinterpret_cast<bf16x4*>(dST + tile_off<256>(j, q_first >> 3) + ((q_first & 7) << 1)) = w;
    }
  }
  if (j < p.Tk) {
    T* dKb = reinterpret_cast<T*>(p.dK) + ((int64_t)b * p.Tk + j) * p.lddk + h * DH;
    T* dVb = reinterpret_cast<T*>(p.dV) + ((int64_t)b * p.Tk + j) * p.lddv + h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      Vec4<T>::store(dKb + 16 * dt + 4 * g, dk[dt] * p.scale);
      Vec4<T>::store(dVb + 16 * dt + 4 * g, dv[dt]);
    }
  }
  IMT_STAMP(p.trace, 2);
  __syncthreads();  // every wave is done with Q / dO; dS^T is complete
  IMT_STAMP(p.trace, 3);
  char* Ks = Qs;
#pragma unroll
  for (int ks = 0; ks < NS; ++ks) *reinterpret_cast<frag_t*>(Ks + tile_off<RB>(k0 + r, 4 * ks + g)) = kf[ks];
  __syncthreads();

  // ---- phase 2: dQ[q][d] = scale * sum_key dS[q][key] K[key][d]
  f32x4 dq[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nks = (p.Tk + 31) / 32;
#pragma unroll 1
  for (int kstep = 0; kstep < nks; ++kstep) {
    const frag_t fa = lds_frag_kstrided_bf16<256>(dST, 32 * kstep, 16 * wave);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) mma16(dq[dt], lds_frag_kstrided_bf16<RB>(Ks, 32 * kstep, 16 * dt), fa);
  }
  const int i = 16 * wave + r;
  if (i < p.Tq) {
    T* dQb = reinterpret_cast<T*>(p.dQ) + ((int64_t)b * p.Tq + i) * p.lddq + h * DH;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) Vec4<T>::store(dQb + 16 * dt + 4 * g, dq[dt] * p.scale);
  }  IMT_STAMP(p.trace, 4);
}

// =========================================================================================== backward: fused, 129-256 keys / queries
// The MASS shapes (BASELINE configs[4]: encoder self-attention 256 x 256, decoder cross-attention 128 x 256) took the
// two-kernel path above: S / dP recomputed twice, Q / K / V / dO read twice, 1.25 ms of a 10.7 ms step.  Same fusion as
// attn_bwd_fused_kernel, re-cut so that it fits 160 KiB of LDS at 256 keys: ONE workgroup (8 waves) per (batch, head),
//   phase 0: K / V rows in registers (KEY on the lane: wave w owns keys 32w .. 32w+31 as two groups of 16), K parked in LDS
//            once for phase 2, delta = rowsum(dO * O) and lse of all queries in LDS;
//   then per tile of 64 QUERIES (its Q / dO rows staged through registers one tile ahead):
//   phase 1: S, P, dP, dS for (64 queries x this wave's 32 keys); dV^T += dO^T P, dK^T += Q^T dS stay in registers over all
//            query tiles; dS^T of the tile goes to LDS as bf16 [key][query];
//   phase 2: QUERY on the lane: dQ[64 x dh] = dS K from LDS (wave = 16 queries x half of dh), stored at once.
// The [key][query] tile is 32 KiB instead of 128: 82 KiB in all.  Nothing is summed across workgroups (no atomics); same
// masks, same dropout draws as every other attention kernel.
template <int DH, bool MASK3D>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused256_kernel(AttnP p) {
  typedef bf16_t T;
  constexpr int RB = DH * 2, NS = RB / 64, NDT = DH / 16, CPR = RB / 16;
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;                         // [256 keys][RB]
  char* Qt = Ks + 256 * RB;                // [64][RB]   current query tile
  char* dOt = Qt + 64 * RB;                // [64][RB]
  char* dST = dOt + 64 * RB;               // [256 keys][64 queries] bf16, 128-B rows
  float* lse_s = reinterpret_cast<float*>(dST + 256 * 128);
  float* delta_s = lse_s + 256;
  uint8_t* qmask_s = reinterpret_cast<uint8_t*>(delta_s + 256);

  const int lid = imt_xcd_block(blockIdx.x, gridDim.x);
  const int b = lid / p.H, h = lid % p.H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)b * p.Tq * p.ldq + h * DH;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)b * p.Tk * p.ldk + h * DH;
  const T* Vb = reinterpret_cast<const T*>(p.V) + (int64_t)b * p.Tk * p.ldv + h * DH;
  const T* dOb = reinterpret_cast<const T*>(p.dO) + (int64_t)b * p.Tq * p.lddo + h * DH;
  const T* Ob = reinterpret_cast<const T*>(p.O) + (int64_t)b * p.Tq * p.ldo + h * DH;
  const int nqt = (p.Tq + 63) / 64;

  // ---- phase 0: every load requested up front, unconditionally (clamped rows / stand-in addresses)
  frag_t kf[2][NS], vf[2][NS];
  bool key_ok[2];
  // one 16-byte chunk of the 64-row Q / dO tile per thread (DH = 32: the first 256 threads), prefetched one tile ahead
  const int st_row = (int)threadIdx.x / CPR, st_c = (int)threadIdx.x % CPR;
  const bool stager = st_row < 64;
  u32x4 vq, vo;
  auto prefetch = [&](int qt) {
    const int tr = min(qt * 64 + min(st_row, 63), p.Tq - 1);
    vq = *reinterpret_cast<const u32x4*>(Qb + (int64_t)tr * p.ldq + st_c * 8);
    vo = *reinterpret_cast<const u32x4*>(dOb + (int64_t)tr * p.lddo + st_c * 8);
  };
  prefetch(0);
  {
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      load_row_frags<T, DH>(kf[kg], Kb, p.ldk, 32 * wave + 16 * kg, p.Tk);
      load_row_frags<T, DH>(vf[kg], Vb, p.ldv, 32 * wave + 16 * kg, p.Tk);
    }
    frag_t of[2][NS], dof[2][NS];
    float lse_v[2];
    uint8_t qmv[2], kmv[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int i = 128 * ps + 16 * wave + r;
      load_row_frags<T, DH>(of[ps], Ob, p.ldo, 128 * ps + 16 * wave, p.Tq);
      load_row_frags<T, DH>(dof[ps], dOb, p.lddo, 128 * ps + 16 * wave, p.Tq);
      lse_v[ps] = p.lse[((int64_t)b * p.H + h) * p.Tq + min(i, p.Tq - 1)];
      const uint8_t* qmp = p.query_mask ? p.query_mask + (int64_t)b * p.Tq + min(i, p.Tq - 1) : reinterpret_cast<const uint8_t*>(Qb);
      qmv[ps] = *qmp;
    }
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int j = 32 * wave + 16 * kg + r;
      const uint8_t* kmp = p.key_mask ? p.key_mask + (int64_t)b * p.Tk + min(j, p.Tk - 1) : reinterpret_cast<const uint8_t*>(Kb);
      kmv[kg] = *kmp;
    }
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int j = 32 * wave + 16 * kg + r;
      key_ok[kg] = (j < p.Tk) ? (p.key_mask ? (kmv[kg] != 0) : true) : false;
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) *reinterpret_cast<frag_t*>(Ks + tile_off<RB>(j, 4 * ks + g)) = kf[kg][ks];  // rows >= Tk: zeros
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int i = 128 * ps + 16 * wave + r;
      float d = 0.f;
#pragma unroll
      for (int ks = 0; ks < NS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)dof[ps][ks][e] * (float)of[ps][ks][e];
      d += __shfl_xor(d, 16, 64);
      d += __shfl_xor(d, 32, 64);
      if (g == 0) {
        delta_s[i] = d;
        lse_s[i] = lse_v[ps];
        qmask_s[i] = (i < p.Tq && p.query_mask) ? qmv[ps] : (uint8_t)1;
      }
    }
  }

  f32x4 dk[2][NDT], dv[2][NDT];
#pragma unroll
  for (int kg = 0; kg < 2; ++kg)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) { dk[kg][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kg][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const uint32_t dkey = dropout_key(p.seed);
  const uint32_t tkp = (uint32_t)((p.Tk + 1) >> 1);
  const int nks = (p.Tk + 31) / 32;
  const int qg = wave & 3, dhalf = wave >> 2;   // phase 2: 16 queries x half of the head dimension per wave
  constexpr int NDH = (NDT + 1) / 2;

#pragma unroll 1
  for (int qt = 0; qt < nqt; ++qt) {
    // the staged tile: rows past Tq are zeros.  (Every wave has left phase 2 of the previous tile -- which reads neither Qt
    // nor dOt -- only after the barrier that closed its phase 1, so these writes cannot overtake a reader.)
    if (stager) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      const bool in = qt * 64 + st_row < p.Tq;
      *reinterpret_cast<u32x4*>(Qt + tile_off<RB>(st_row, st_c)) = in ? vq : z;
      *reinterpret_cast<u32x4*>(dOt + tile_off<RB>(st_row, st_c)) = in ? vo : z;
    }
    __syncthreads();  // tile staged (first trip: K, delta, lse too); dS^T of the previous tile fully consumed
    if (qt + 1 < nqt) prefetch(qt + 1);

    // ---- phase 1
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int j = 32 * wave + 16 * kg + r;  // this lane's key
      f32x4 s[4], dp[4], pd[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        s[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dp[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
          mma16(s[mt], lds_frag_kcontig<T, RB>(Qt, 16 * mt, 4 * ks), kf[kg][ks]);
          mma16(dp[mt], lds_frag_kcontig<T, RB>(dOt, 16 * mt, 4 * ks), vf[kg][ks]);
        }
      }
      const uint32_t pair_base = (uint32_t)(b * p.H + h) * (uint32_t)p.Tq * tkp + (uint32_t)(j >> 1);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int i0 = qt * 64 + 16 * mt + 4 * g;  // this lane's 4 consecutive queries
        const f32x4 lse4 = *reinterpret_cast<const f32x4*>(lse_s + i0), del4 = *reinterpret_cast<const f32x4*>(delta_s + i0);
        const uint32_t qm4 = *reinterpret_cast<const uint32_t*>(qmask_s + i0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int i = i0 + e;
          float v = s[mt][e] * p.scale + (mask_ok<MASK3D>(p, b, i, j, key_ok[kg], ((qm4 >> (8 * e)) & 0xffu) != 0) ? 0.f : -10000.0f);
          float pv = (i < p.Tq && j < p.Tk) ? __expf(v - lse4[e]) : 0.f;
          float dpv = dp[mt][e], pdv = pv;
          if (p.drop_thresh) {  // (host guarantees B*H*Tq*Tk < 2^32 for this kernel)
            const bool keep = attn_drop_keep(attn_drop_word(dkey, pair_base + (uint32_t)i * tkp), j, p.drop_thresh);
            dpv = keep ? dpv * p.inv_keep : 0.f;
            pdv = keep ? pv * p.inv_keep : 0.f;
          }
          pd[mt][e] = pdv;
          s[mt][e] = pv * (dpv - del4[e]);  // dS[i][j]
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const frag_t pf = acc_pair_to_frag(pd[2 * u], pd[2 * u + 1]);
        const frag_t dsf = acc_pair_to_frag(s[2 * u], s[2 * u + 1]);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          mma16(dv[kg][dt], lds_frag_kperm_bf16<RB>(dOt, 32 * u, 16 * dt), pf);   // dV^T[d][key] += dO^T[d][q] P[q][key]
          mma16(dk[kg][dt], lds_frag_kperm_bf16<RB>(Qt, 32 * u, 16 * dt), dsf);   // dK^T[d][key] += Q^T[d][q] dS[q][key]
        }
      }
      // dS^T[key j][local queries 16mt + 4g .. +3] (bf16, 8 bytes)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int ql = 16 * mt + 4 * g;
        bf16x4 w = {(bf16_t)s[mt][0], (bf16_t)s[mt][1], (bf16_t)s[mt][2], (bf16_t)s[mt][3]};
        *reinterpret_cast<bf16x4*>(dST + tile_off<128>(j, ql >> 3) + ((ql & 7) << 1)) = w;
      }
    }
    __syncthreads();  // dS^T of this tile complete

    // ---- phase 2: dQ[q][d] = scale * sum_key dS[q][key] K[key][d]
    f32x4 dq[NDH];
#pragma unroll
    for (int t = 0; t < NDH; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (dhalf * NDH < NDT) {  // (uniform per wave; DH = 32 with NDT = 2: both halves hold one tile)
#pragma unroll 2
      for (int kstep = 0; kstep < nks; ++kstep) {
        const frag_t fa = lds_frag_kstrided_bf16<128>(dST, 32 * kstep, 16 * qg);
#pragma unroll
        for (int t = 0; t < NDH; ++t) mma16(dq[t], lds_frag_kstrided_bf16<RB>(Ks, 32 * kstep, 16 * (dhalf * NDH + t)), fa);
      }
      const int i = qt * 64 + 16 * qg + r;
      if (i < p.Tq) {
        T* dQb = reinterpret_cast<T*>(p.dQ) + ((int64_t)b * p.Tq + i) * p.lddq + h * DH;
#pragma unroll
        for (int t = 0; t < NDH; ++t) Vec4<T>::store(dQb + 16 * (dhalf * NDH + t) + 4 * g, dq[t] * p.scale);
      }
    }
  }
#pragma unroll
  for (int kg = 0; kg < 2; ++kg) {
    const int j = 32 * wave + 16 * kg + r;
    if (j < p.Tk) {
      T* dKb = reinterpret_cast<T*>(p.dK) + ((int64_t)b * p.Tk + j) * p.lddk + h * DH;
      T* dVb = reinterpret_cast<T*>(p.dV) + ((int64_t)b * p.Tk + j) * p.lddv + h * DH;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        Vec4<T>::store(dKb + 16 * dt + 4 * g, dk[kg][dt] * p.scale);
        Vec4<T>::store(dVb + 16 * dt + 4 * g, dv[kg][dt]);
      }
    }
  }
}

// =========================================================================================== forward: q|k|v projection + attention
// Self-attention of a short sequence (T <= 128, head_dim 64, bf16) with the q|k|v projection of BertSelfAttention
// (src/bert_seq2seq.py:84-90,139-143 -> HF BertSelfAttention.query / key / value) INSIDE the attention launch: one workgroup
// (8 waves) per (batch element, PAIR of heads) computes its 128 x 384 slice of x W^T + b -- q, k and v of its two heads, the
// 128 rows being the batch element's tokens -- with the 256-tile kernel's loop (LDS-DMA of K tiles by all waves, two 64-KiB
// stages, one barrier per K tile), stores it (the backward reads q|k|v) and parks it in LDS as the Q / K / V tiles of
// attn_fwd_short_kernel, whose arithmetic then runs unchanged for the two heads.  Against the projection launch + attention
// launch it saves one launch boundary, the 24-MB re-read of q|k|v and the second cold start.  Same k order per accumulator as
// every GEMM kernel and the same attention code: results are bit-identical to the two-launch path.
struct QkvP { const void* x; int64_t ldx; int64_t x_bytes; const void* w; const void* bias; int d_model; };

// one wave's four 1-KiB pieces of a [128 rows][64 k] K-contiguous bf16 tile (the Dma struct of gemm.hip)
struct QkvDma {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff[4];
  IMT_DEVICE void init(const bf16_t* base, int64_t ld, int64_t valid_bytes, int row0, int sub) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, (int)valid_bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = 64 * (4 * i + sub) + lane;
      const int tr = q >> 3, pc = q & 7;
      const int c = pc ^ swz<128>(tr);
      voff[i] = (int)((((int64_t)(row0 + tr)) * ld + c * 8) * 2);
    }
  }
  IMT_DEVICE void issue(char* tile, int sub, int t) const {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(tile + (4 * i + sub) * 1024), 16,
                                               voff[i] + t * 128, 0, 0, 0);
  }
};

template <bool UNUSED = false>
__global__ __launch_bounds__(512, 2) void attn_qkv_fwd_kernel(AttnP p, QkvP g) {
  typedef bf16_t T;
  constexpr int DH = 64, RB = 128, NS = 2, NDT = 4;
  typedef Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K loop: 2 stages x (x tile | Wq | Wk | Wv tiles of 16 KiB)
  const int hp_n = p.H >> 1;
  const int lid = imt_xcd_block(blockIdx.x, gridDim.x);
  const int b = lid / hp_n, hp = lid % hp_n;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, gq = lane >> 4;
  const int T_ = p.Tq, d = g.d_model;
  const int nt = d >> 6;  // K tiles of 64 (host: d % 64 == 0)

  // ---- phase A: [128 tokens] x [384 = q | k | v of two heads] = x_b W_sel^T
  QkvDma d0, d1;
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(g.x) + (int64_t)b * T_ * g.ldx;
  const bf16_t* wq = reinterpret_cast<const bf16_t*>(g.w);
  const int sub = wave & 3;
  if (wave < 4) {  // waves 0-3 stream the x tile and the Wq tile, waves 4-7 the Wk and Wv tiles
    d0.init(xb, g.ldx, g.x_bytes - (int64_t)b * T_ * g.ldx * 2, 0, sub);
    d1.init(wq, d, (int64_t)3 * d * d * 2, hp * 128, sub);
  } else {
    d0.init(wq, d, (int64_t)3 * d * d * 2, d + hp * 128, sub);
    d1.init(wq, d, (int64_t)3 * d * d * 2, 2 * d + hp * 128, sub);
  }
  const int t0 = wave < 4 ? 0 : 2;  // first of this wave's two sub-tiles within a stage
  auto issue = [&](int slot, int t) {
    d0.issue(smem + slot * 65536 + t0 * 16384, sub, t);
    d1.issue(smem + slot * 65536 + (t0 + 1) * 16384, sub, t);
  };
  const int wm = (wave >> 2) * 64, c0 = (wave & 3) * 96;  // this wave's 64 rows x 96 columns
  f32x4 acc[4][6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (t + 1 < nt) issue((t + 1) & 1, t + 1);
    const char* st = smem + (t & 1) * 65536;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      frag_t fa[4], fb[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int c = c0 + 16 * j;
        fb[j] = lds_frag_kcontig<T, RB>(st + (1 + (c >> 7)) * 16384, c & 127, 4 * s);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = lds_frag_kcontig<T, RB>(st, wm + 16 * i, 4 * s);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) mma16(acc[i][j], fb[j], fa[i]);  // C^T tile: rows <- n, cols <- m
    }
  }
  __syncthreads();  // every wave is done with the last K tile: the stages become the Q / K / V tiles

  // ---- phase B: + bias, store q|k|v (the backward reads them), park them as [which][head][128 rows][64] bf16 tiles
  char* tiles = smem;                       // tile (which, hh) at ((which * 2 + hh) * 16384)
  uint8_t* kmask_s = reinterpret_cast<uint8_t*>(smem + 6 * 16384);
  {
    const bf16_t* bias = reinterpret_cast<const bf16_t*>(g.bias);
    bf16_t* outs[3] = {reinterpret_cast<bf16_t*>(const_cast<void*>(p.Q)), reinterpret_cast<bf16_t*>(const_cast<void*>(p.K)),
                       reinterpret_cast<bf16_t*>(const_cast<void*>(p.V))};
    const int64_t lds_[3] = {p.ldq, p.ldk, p.ldv};
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int c = c0 + 16 * j + 4 * gq;        // column of the 384: which = c >> 7, head = (c >> 6) & 1, dim = c & 63
      const int which = c >> 7, hh = (c >> 6) & 1, dim = c & 63;
      const int gcol = which * d + hp * 128 + (c & 127);
      const f32x4 bv = bias ? Vec4<T>::load(bias + gcol) : f32x4{0.f, 0.f, 0.f, 0.f};
      bf16_t* ob = outs[which] + (int64_t)b * T_ * lds_[which] + hp * 128 + (c & 127);
      char* tl = tiles + (which * 2 + hh) * 16384;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = wm + 16 * i + r;
        const f32x4 v = acc[i][j] * 1.0f + bv;
        bf16x4 w4 = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        if (m >= T_) w4 = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};  // rows of the next batch element / past the end
        else *reinterpret_cast<bf16x4*>(ob + (int64_t)m * lds_[which]) = w4;
        *reinterpret_cast<bf16x4*>(tl + tile_off<RB>(m, dim >> 3) + ((dim & 7) << 1)) = w4;
      }
    }
    if (threadIdx.x < 128) {
      const int j = threadIdx.x;
      kmask_s[j] = (j < p.Tk) ? (p.key_mask ? p.key_mask[(int64_t)b * p.Tk + j] : (uint8_t)1) : (uint8_t)0;
    }
  }
  const int q0 = wave * 16, i = q0 + r;
  const uint8_t* qmp = p.query_mask ? p.query_mask + (int64_t)b * p.Tq + min(i, p.Tq - 1) : reinterpret_cast<const uint8_t*>(g.w);
  const uint8_t qmv = *qmp;
  const bool query_ok = (p.query_mask && i < p.Tq) ? (qmv != 0) : true;
  __syncthreads();
  if (q0 >= p.Tq) return;  // whole wave past the last query (no barrier follows)

  // ---- phase C: attn_fwd_short_kernel's arithmetic for the two heads
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int h = 2 * hp + hh;
    const char* Qs = tiles + (0 * 2 + hh) * 16384;
    const char* Ks = tiles + (1 * 2 + hh) * 16384;
    const char* Vs = tiles + (2 * 2 + hh) * 16384;
    frag_t qf[NS];
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) qf[ks] = lds_frag_kcontig<T, RB>(Qs, q0, 4 * ks);
    f32x4 s[8];
    float tmax = -INFINITY;
#pragma unroll
    for (int nt8 = 0; nt8 < 8; ++nt8) {
      s[nt8] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) mma16(s[nt8], lds_frag_kcontig<T, RB>(Ks, 16 * nt8, 4 * ks), qf[ks]);
      const uint32_t km4 = *reinterpret_cast<const uint32_t*>(kmask_s + 16 * nt8 + 4 * gq);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = 16 * nt8 + 4 * gq + e;
        float v = s[nt8][e] * p.scale + (mask_ok<false>(p, b, i, j, ((km4 >> (8 * e)) & 0xffu) != 0, query_ok) ? 0.f : -10000.0f);
        if (j >= p.Tk) v = -INFINITY;
        s[nt8][e] = v;
        tmax = fmaxf(tmax, v);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    float psum = 0.f;
#pragma unroll
    for (int nt8 = 0; nt8 < 8; ++nt8)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pv = __expf(s[nt8][e] - tmax);
        psum += pv;
        s[nt8][e] = pv;
      }
    if (p.drop_thresh) {
      const uint32_t key = dropout_key(p.seed);
      const uint32_t prow = (uint32_t)attn_pair_row(p, b, h, i) + (uint32_t)(2 * gq);
#pragma unroll
      for (int nt8 = 0; nt8 < 8; ++nt8)
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          const uint32_t w = attn_drop_word(key, prow + 8 * nt8 + e2);
          s[nt8][2 * e2] = (w & 0xffffu) >= p.drop_thresh ? s[nt8][2 * e2] * p.inv_keep : 0.f;
          s[nt8][2 * e2 + 1] = (w >> 16) >= p.drop_thresh ? s[nt8][2 * e2 + 1] * p.inv_keep : 0.f;
        }
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    f32x4 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const frag_t pf = acc_pair_to_frag(s[2 * u], s[2 * u + 1]);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) mma16(o[dt], lds_frag_kperm_bf16<RB>(Vs, 32 * u, 16 * dt), pf);
    }
    const float inv_l = 1.0f / psum;
    if (i < p.Tq) {
      T* Ob = reinterpret_cast<T*>(p.O) + ((int64_t)b * p.Tq + i) * p.ldo + h * DH;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) Vec4<T>::store(Ob + 16 * dt + 4 * gq, o[dt] * inv_l);
      if (gq == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.Tq + i] = tmax + __logf(psum);
    }
  }
}

int check_args(const imt_attn_args* a, bool bwd) {
  IMT_CHECK_ARG(a != nullptr, "attention: null args");
  IMT_CHECK_ARG(a->dtype == IMT_F32 || a->dtype == IMT_BF16, "attention: bad dtype");
  IMT_CHECK_ARG(a->head_dim == 32 || a->head_dim == 64, "attention: head_dim %d unsupported (32 or 64)", a->head_dim);
  IMT_CHECK_ARG(a->B > 0 && a->H > 0 && a->Tq > 0 && a->Tk > 0, "attention: bad dims");
  const int al = (a->dtype == IMT_BF16) ? 8 : 4;
  IMT_CHECK_ARG(a->Q && a->K && a->V && a->O, "attention: null tensor");
  IMT_CHECK_ARG(a->ldq % al == 0 && a->ldk % al == 0 && a->ldv % al == 0 && a->ldo % al == 0, "attention: ld must be 16-B multiples");
  IMT_CHECK_ARG((((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V | (uintptr_t)a->O) & 15) == 0, "attention: 16-B alignment");
  if (bwd) {
    IMT_CHECK_ARG(a->dO && a->dQ && a->dK && a->dV && a->lse && a->delta, "attention_bwd: null tensor");
    IMT_CHECK_ARG(a->lddo % al == 0 && a->lddq % al == 0 && a->lddk % al == 0 && a->lddv % al == 0, "attention_bwd: ld alignment");
    IMT_CHECK_ARG((((uintptr_t)a->dO | (uintptr_t)a->dQ | (uintptr_t)a->dK | (uintptr_t)a->dV) & 15) == 0, "attention_bwd: 16-B alignment");
  }
  return IMT_OK;
}

AttnP make_params(const imt_attn_args* a) {
  AttnP p;
  p.B = a->B; p.H = a->H; p.Tq = a->Tq; p.Tk = a->Tk;
  p.Q = a->Q; p.ldq = a->ldq; p.K = a->K; p.ldk = a->ldk; p.V = a->V; p.ldv = a->ldv; p.O = a->O; p.ldo = a->ldo;
  p.lse = a->lse; p.key_mask = a->key_mask; p.query_mask = a->query_mask; p.mask3d = a->mask3d; p.causal = a->causal;
  p.scale = a->scale;
  p.drop_thresh = dropout_thresh(a->dropout_p);
  p.inv_keep = a->dropout_p > 0.f ? 1.f / (1.f - a->dropout_p) : 1.f;
  p.seed = a->dropout_seed;
  p.dO = a->dO; p.lddo = a->lddo; p.dQ = a->dQ; p.lddq = a->lddq; p.dK = a->dK; p.lddk = a->lddk; p.dV = a->dV; p.lddv = a->lddv; p.trace = nullptr;
  p.delta = a->delta;
  return p;
}

template <typename T, int DH> int fwd_launch(const AttnP& p, hipStream_t st) {
  dim3 grid(imt_cdiv(p.Tq, 64) * p.H * p.B);
  const double work = (double)p.B * p.H * p.Tq * p.Tk * DH;
  const double io = ((double)p.B * p.H * DH * sizeof(T)) * (2.0 * p.Tq + 2.0 * p.Tk);
  ImtProfScope prof(sizeof(T) == 2 ? "attn_fwd_bf16" : "attn_fwd_f32", 4.0 * work, io, st);
  if (p.mask3d) hipLaunchKernelGGL((attn_fwd_kernel<T, DH, true>), grid, dim3(256), 0, st, p);
  else          hipLaunchKernelGGL((attn_fwd_kernel<T, DH, false>), grid, dim3(256), 0, st, p);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}
template <int DH, bool MASK3D> int bwd_fused_launch(const AttnP& p, hipStream_t st) {
  const double work = (double)p.B * p.H * p.Tq * p.Tk * DH;
  const double io = ((double)p.B * p.H * DH * 2.0) * (4.0 * p.Tq + 4.0 * p.Tk);
  const int lds = 2 * 128 * DH * 2 + 128 * 256 + 128 * 4 * 2 + 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_kernel<DH, MASK3D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  ImtProfScope prof("attn_bwd_fused_bf16", 10.0 * work, io, st);
  ImtTrace tr("attn_bwd", p.B * p.H, st);
  AttnP pt = p;
  pt.trace = tr.dev;
  hipLaunchKernelGGL((attn_bwd_fused_kernel<DH, MASK3D>), dim3(p.B * p.H), dim3(512), lds, st, pt);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}
template <int DH, bool MASK3D> int bwd_fused256_launch(const AttnP& p, hipStream_t st) {
  const double work = (double)p.B * p.H * p.Tq * p.Tk * DH;
  const double io = ((double)p.B * p.H * DH * 2.0) * (4.0 * p.Tq + 4.0 * p.Tk);
  const int lds = 256 * DH * 2 + 2 * 64 * DH * 2 + 256 * 128 + 256 * 4 * 2 + 256;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused256_kernel<DH, MASK3D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  ImtProfScope prof("attn_bwd_fused256_bf16", 10.0 * work, io, st);
  hipLaunchKernelGGL((attn_bwd_fused256_kernel<DH, MASK3D>), dim3(p.B * p.H), dim3(512), lds, st, p);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}
template <typename T, int DH> int bwd_launch(const AttnP& p, hipStream_t st) {
  const double work = (double)p.B * p.H * p.Tq * p.Tk * DH;
  const double io = ((double)p.B * p.H * DH * sizeof(T)) * (2.0 * p.Tq + 2.0 * p.Tk);
  {
    ImtProfScope prof(sizeof(T) == 2 ? "attn_bwd_dq_bf16" : "attn_bwd_dq_f32", 6.0 * work, io * 1.5, st);
    if (p.mask3d) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DH, true>), dim3(imt_cdiv(p.Tq, 64) * p.H * p.B), dim3(256), 0, st, p);
    else          hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DH, false>), dim3(imt_cdiv(p.Tq, 64) * p.H * p.B), dim3(256), 0, st, p);
    IMT_CHECK_LAUNCH();
  }
  {
    ImtProfScope prof(sizeof(T) == 2 ? "attn_bwd_dkdv_bf16" : "attn_bwd_dkdv_f32", 8.0 * work, io * 1.5, st);
    if (p.mask3d) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, DH, true>), dim3(imt_cdiv(p.Tk, 64) * p.H * p.B), dim3(256), 0, st, p);
    else          hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, DH, false>), dim3(imt_cdiv(p.Tk, 64) * p.H * p.B), dim3(256), 0, st, p);
    IMT_CHECK_LAUNCH();
  }
  return IMT_OK;
}

}  // namespace

extern "C" int imt_attention_fwd(const imt_attn_args* a, void* stream) {
  int rc = check_args(a, false);
  if (rc) return rc;
  const AttnP p = make_params(a);
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == IMT_F32) return a->head_dim == 32 ? fwd_launch<float, 32>(p, st) : fwd_launch<float, 64>(p, st);
  const bool idx32 = (double)a->B * a->H * a->Tq * a->Tk < 4294967296.0;  // 32-bit dropout element indices
  if (a->Tq <= 128 && a->Tk <= 128 && a->Tq > 64 && idx32 && !getenv("IMT_ATTN_NO_SHORT_FWD")) {  // one workgroup per (batch, head)
    const double work = (double)p.B * p.H * p.Tq * p.Tk * a->head_dim;
    ImtProfScope prof("attn_fwd_bf16", 4.0 * work, ((double)p.B * p.H * a->head_dim * 2.0) * (2.0 * p.Tq + 2.0 * p.Tk), st);
    ImtTrace tr("attn_fwd", p.B * p.H, st);
    AttnP pt = p;
    pt.trace = tr.dev;
    if (a->head_dim == 32) {
      if (p.mask3d) hipLaunchKernelGGL((attn_fwd_short_kernel<32, true>), dim3(p.B * p.H), dim3(512), 0, st, pt);
      else          hipLaunchKernelGGL((attn_fwd_short_kernel<32, false>), dim3(p.B * p.H), dim3(512), 0, st, pt);
    } else {
      if (p.mask3d) hipLaunchKernelGGL((attn_fwd_short_kernel<64, true>), dim3(p.B * p.H), dim3(512), 0, st, pt);
      else          hipLaunchKernelGGL((attn_fwd_short_kernel<64, false>), dim3(p.B * p.H), dim3(512), 0, st, pt);
    }
    IMT_CHECK_LAUNCH();
    return IMT_OK;
  }
  return a->head_dim == 32 ? fwd_launch<bf16_t, 32>(p, st) : fwd_launch<bf16_t, 64>(p, st);
}

extern "C" int imt_attention_bwd(const imt_attn_args* a, void* stream) {
  int rc = check_args(a, true);
  if (rc) return rc;
  const AttnP p = make_params(a);
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == IMT_F32) return a->head_dim == 32 ? bwd_launch<float, 32>(p, st) : bwd_launch<float, 64>(p, st);
  const bool idx32 = (double)a->B * a->H * a->Tq * a->Tk < 4294967296.0;  // 32-bit dropout element indices
  if (a->Tq <= 128 && a->Tk <= 128 && idx32 && !getenv("IMT_ATTN_NO_FUSED_BWD")) {  // short sequences: one fused kernel
    if (p.mask3d) return a->head_dim == 32 ? bwd_fused_launch<32, true>(p, st) : bwd_fused_launch<64, true>(p, st);
    return a->head_dim == 32 ? bwd_fused_launch<32, false>(p, st) : bwd_fused_launch<64, false>(p, st);
  }
  // 129 .. 256 keys / queries (MASS: encoder self-attention at 256 tokens, the decoder's cross-attention over them)
  if (a->Tq <= 256 && a->Tk <= 256 && idx32 && !getenv("IMT_ATTN_NO_FUSED_BWD") && !getenv("IMT_ATTN_NO_FUSED256")) {
    if (p.mask3d) return a->head_dim == 32 ? bwd_fused256_launch<32, true>(p, st) : bwd_fused256_launch<64, true>(p, st);
    return a->head_dim == 32 ? bwd_fused256_launch<32, false>(p, st) : bwd_fused256_launch<64, false>(p, st);
  }
  return a->head_dim == 32 ? bwd_launch<bf16_t, 32>(p, st) : bwd_launch<bf16_t, 64>(p, st);
}

// q|k|v projection + self-attention forward in one launch (attn_qkv_fwd_kernel).  a: the attention arguments with Q / K / V
// pointing at the OUTPUT views of the projection (element (b, t, h, e) as for imt_attention_fwd); x [B*T, d_model] (ldx), w
// [3 d_model, d_model] row-major (q rows first, then k, then v: the runtime's fused projection weight), bias [3 d_model] or NULL.
extern "C" int imt_attention_qkv_fwd_supported(int dtype, int head_dim, int H, int Tq, int Tk, int d_model, int has_mask3d) {
  return dtype == IMT_BF16 && head_dim == 64 && H % 2 == 0 && H * head_dim == d_model && Tq == Tk && Tq > 64 && Tq <= 128 &&
         d_model % 64 == 0 && d_model >= 128 && !has_mask3d;
}
extern "C" int imt_attention_qkv_fwd(const imt_attn_args* a, const void* x, int64_t ldx, const void* w, const void* bias, int d_model,
                                     void* stream) {
  int rc = check_args(a, false);
  if (rc) return rc;
  IMT_CHECK_ARG(x && w && ldx % 8 == 0 && (((uintptr_t)x | (uintptr_t)w) & 15) == 0, "attention_qkv_fwd: x / w missing or misaligned");
  IMT_CHECK_ARG(imt_attention_qkv_fwd_supported(a->dtype, a->head_dim, a->H, a->Tq, a->Tk, d_model, a->mask3d != nullptr),
                "attention_qkv_fwd: unsupported shape (bf16, head_dim 64, even heads, 64 < T <= 128, self-attention, no 3-D mask)");
  IMT_CHECK_ARG((double)a->B * a->H * a->Tq * a->Tk < 4294967296.0 && (int64_t)a->B * a->Tq * ldx * 2 < (1ll << 31) &&
                (int64_t)3 * d_model * d_model * 2 < (1ll << 31), "attention_qkv_fwd: operand too large for 32-bit offsets");
  const AttnP p = make_params(a);
  QkvP g;
  g.x = x; g.ldx = ldx; g.x_bytes = ((int64_t)(a->B * a->Tq - 1) * ldx + d_model) * 2; g.w = w; g.bias = bias; g.d_model = d_model;
  hipStream_t st = (hipStream_t)stream;
  const int lds = 2 * 65536;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_qkv_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const double work = (double)p.B * p.H * p.Tq * p.Tk * a->head_dim;
  ImtProfScope prof("attn_qkv_fwd_bf16", 4.0 * work + 2.0 * p.B * p.Tq * 3.0 * d_model * d_model,
                    ((double)p.B * p.Tq * d_model * 2.0) * 5.0 + 6.0 * d_model * d_model, st);
  hipLaunchKernelGGL((attn_qkv_fwd_kernel<false>), dim3(p.B * (p.H / 2)), dim3(512), lds, st, p, g);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}
