// 16x16 MFMA tile primitives for gfx950, written once for fp32 (v_mfma_f32_16x16x4_f32, exact fmaf chain)
// and bf16 (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
//
// Conventions (cdna_hip_programming.md section 3):
//   D = A*B + C, per wave.  lane l: r = l & 15, g = l >> 4.
//   A fragment: lane holds A[row r][k-slots of group g]; B fragment: lane holds B[k-slots of group g][col r].
//   D fragment: lane holds D[row 4g+e][col r], e = 0..3.
//
// A "fragment" here always covers ONE 16-byte chunk of the operand's K extent per lane:
//   bf16: 8 consecutive k  (one 16x16x32 MFMA per fragment pair, k = 8g + j)
//   fp32: 4 consecutive k  (four 16x16x4 MFMAs per fragment pair; MFMA j uses element j, k = 4g + j)
// so a "k-step" covers 64 bytes of K (32 bf16 / 16 fp32) for both types and the LDS byte geometry of a
// tile is identical for both types.
#pragma once
#include "common.hpp"

template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x4 type; static constexpr int KSTEP = 16; static constexpr int EPC = 4; };
template <> struct Frag<bf16_t> { typedef bf16x8 type; static constexpr int KSTEP = 32; static constexpr int EPC = 8; };

// acc += A(frag a) * B(frag b)
IMT_DEVICE void mma16(f32x4& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
}
IMT_DEVICE void mma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

// ---------------------------------------------------------------- swizzled LDS tiles
// A tile is [rows][RB bytes per row], RB a power of two >= 64.  16-byte chunk c of row r is stored at
// chunk (c ^ swz(r)); swz spreads the rows that share a 256-byte bank row over distinct 16-byte slots so
// that ds_read_b128 of "same chunk, 16 consecutive rows" is conflict-free (T2 in the guide).
template <int RB> IMT_DEVICE int swz(int row) {
  constexpr int CPR = RB / 16;                       // chunks per row
  constexpr int RPB = (RB >= 256) ? 1 : (256 / RB);  // rows per 256-B bank row
  constexpr int MASK = (CPR < 16 ? CPR : 16) - 1;
  return (row / RPB) & MASK;
}
template <int RB> IMT_DEVICE int tile_off(int row, int chunk) { return row * RB + ((chunk ^ swz<RB>(row)) << 4); }

// ---- K-contiguous operand: tile[row][k], fragment = one 16-B chunk at (row0 + r, chunk)
template <typename T, int RB>
IMT_DEVICE typename Frag<T>::type lds_frag_kcontig(const char* tile, int row0, int chunk0) {
  const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
  return *reinterpret_cast<const typename Frag<T>::type*>(tile + tile_off<RB>(row0 + r, chunk0 + g));
}

// ---- K-strided operand: tile[k][col] (col contiguous).  Fragment for 16 columns col0..col0+15 (col0 % 16 == 0)
// and the k-step whose first tile row is krow0: lane (r,g) needs tile[krow0 + KPG*g + j][col0 + r].
template <int RB> IMT_DEVICE f32x4 lds_frag_kstrided_f32(const char* tile, int krow0, int col0) {
  const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
  const int col = col0 + r;
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int kr = krow0 + 4 * g + j;
    v[j] = *reinterpret_cast<const float*>(tile + tile_off<RB>(kr, col >> 2) + ((col & 3) << 2));
  }
  return v;
}

#ifndef IMT_NO_TR_READ
// ds_read_b64_tr_b16 (T10): within a 16-lane group, lane 4q+p supplies the address of row q, columns
// 4p..4p+3 of a 4x16 block; lane i receives column i, row e in element e.  EXEC must be all ones.
// (Verified on MI355X with tools/probe_tr.hip / tools/probe_frag.hip.)
typedef __attribute__((ext_vector_type(8))) short s16x8;
IMT_DEVICE s16x4 tr_read16(const char* addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<char*>(addr)));
}
IMT_DEVICE bf16x8 join_tr(s16x4 lo, s16x4 hi) {
  // whole-vector bit cast (per-element bit_cast of vector lanes miscompiled: every lane got element 0)
  const s16x8 w = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, w);
}
template <int RB> IMT_DEVICE bf16x8 lds_frag_kstrided_bf16(const char* tile, int krow0, int col0) {
  const int l = threadIdx.x & 63, i = l & 15, g = l >> 4, q = i >> 2, p = i & 3;
  const int col = col0 + 4 * p;
  s16x4 t[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int kr = krow0 + 8 * g + 4 * h + q;
    t[h] = tr_read16(tile + tile_off<RB>(kr, col >> 3) + ((col & 7) << 1));
  }
  return join_tr(t[0], t[1]);
}
#else
// Debug fallback: 8 scalar 2-byte LDS reads (no transposed read); used to cross-check the tr path.
template <int RB> IMT_DEVICE bf16x8 lds_frag_kstrided_bf16(const char* tile, int krow0, int col0) {
  const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
  const int col = col0 + r;
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kr = krow0 + 8 * g + j;
    v[j] = *reinterpret_cast<const bf16_t*>(tile + tile_off<RB>(kr, col >> 3) + ((col & 7) << 1));
  }
  return v;
}
#endif

template <typename T, int RB> struct KStrided;
template <int RB> struct KStrided<float, RB> {
  static IMT_DEVICE f32x4 load(const char* tile, int krow0, int col0) { return lds_frag_kstrided_f32<RB>(tile, krow0, col0); }
};
template <int RB> struct KStrided<bf16_t, RB> {
  static IMT_DEVICE bf16x8 load(const char* tile, int krow0, int col0) { return lds_frag_kstrided_bf16<RB>(tile, krow0, col0); }
};

// ---- permuted-K variant used when the OTHER operand is an MFMA accumulator (attention P, dS):
// an accumulator tile-pair (two 16x16 D tiles t0,t1 stacked along the row index) gives lane (r,g) the rows
// 16*t + 4g + e.  As an operand its k-slot order is therefore
//   bf16 : slot j (0..7) -> tile (j >> 2), row 4g + (j & 3)         (one MFMA per tile PAIR)
//   fp32 : slot e (0..3) -> row 4g + e of one tile                  (one fragment per tile)
// The matching K-strided fragment of the LDS operand reads exactly those rows.
template <int RB> IMT_DEVICE f32x4 lds_frag_kperm_f32(const char* tile, int krow_tile0, int col0) {
  // rows krow_tile0 + 4g + e : identical to the plain k-strided f32 fragment (KPG = 4)
  return lds_frag_kstrided_f32<RB>(tile, krow_tile0, col0);
}
#ifndef IMT_NO_TR_READ
template <int RB> IMT_DEVICE bf16x8 lds_frag_kperm_bf16(const char* tile, int krow_tile0, int col0) {
  const int l = threadIdx.x & 63, i = l & 15, g = l >> 4, q = i >> 2, p = i & 3;
  const int col = col0 + 4 * p;
  s16x4 t[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int kr = krow_tile0 + 16 * h + 4 * g + q;
    t[h] = tr_read16(tile + tile_off<RB>(kr, col >> 3) + ((col & 7) << 1));
  }
  return join_tr(t[0], t[1]);
}
#else
template <int RB> IMT_DEVICE bf16x8 lds_frag_kperm_bf16(const char* tile, int krow_tile0, int col0) {
  const int l = threadIdx.x & 63, r = l & 15, g = l >> 4;
  const int col = col0 + r;
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kr = krow_tile0 + 16 * (j >> 2) + 4 * g + (j & 3);
    v[j] = *reinterpret_cast<const bf16_t*>(tile + tile_off<RB>(kr, col >> 3) + ((col & 7) << 1));
  }
  return v;
}
#endif

// Pack accumulator tiles into an operand fragment (see above).
IMT_DEVICE bf16x8 acc_pair_to_frag(const f32x4& t0, const f32x4& t1) {
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = (bf16_t)t0[e]; v[4 + e] = (bf16_t)t1[e]; }
  return v;
}
