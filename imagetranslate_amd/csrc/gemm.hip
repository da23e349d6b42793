// MFMA GEMM for every nn.Linear forward/backward on the path (see include/imt_hip.h, imt_gemm).
//
// Tile: 128 x 128 output per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles),
// K advanced 128 BYTES per tile row (64 bf16 / 32 fp32) so the LDS geometry is the same for both types.
// K-contiguous operands are read back from XOR-swizzled LDS with ds_read_b128; K-strided operands (the
// N-contiguous B of NN, both operands of TN) with the gfx950 transposed LDS read ds_read_b64_tr_b16 (bf16) or
// scalar reads (fp32) -- no operand is ever transposed through HBM.  The product is accumulated as C^T tiles
// (mfma(Bfrag, Afrag)) so each lane owns 4 consecutive n of one m: bias / residual / aux accesses and the C
// store are 8- or 16-byte vectors.
//
// Five main loops share the tile product (compute_tile) and the LDS-restaged epilogue; imt_gemm picks one per shape
// from measurements (profiles/r01_v5_gemm_shapes.txt, profiles/r01_gemm_epilogue_study.txt):
//   gemm_ws_kernel         : persistent, wave-specialised -- 4 MFMA waves + 4 LDS-DMA producer waves (buffer_load ... lds,
//                            16 B/lane, swizzle on the SOURCE address, 3-stage ring, one raw s_barrier per K tile, counted
//                            vmcnt on the producers); on a workgroup's last tile the idle producers share the epilogue.
//                            K a whole number of tiles and (<= one tile per CU or K >= 1024).
//   gemm_grouped_tn_kernel : the same split for ALL weight-gradient products of a layer in one launch (full K per tile,
//                            fp32 accumulate, fused bias gradients).
//   gemm_sb_kernel         : one 32-KiB LDS buffer + register prefetch, three workgroups per CU overlap each other's
//                            barriers and epilogues: many short-K tiles.
//   gemm_kernel            : register-staged double buffer with predicated (zero-filling) loads: ragged / tiny K, split-K.
//   gemm_pipe_kernel       : LDS-DMA ring issued by the MFMA waves themselves (superseded by gemm_ws_kernel; kept as a
//                            cross-check variant for the tests).
#include <stdlib.h>
#include "mma.hpp"

// ---- build-time tuning switches (tools/build_variant.sh builds a second library with other values for A/B runs)
#ifndef IMT_WS_NST
#define IMT_WS_NST 3
#endif
#ifndef IMT_LN_TICKET
#define IMT_LN_TICKET 0  // 1: imt_gemm's ln_out normalises finished row blocks inside the persistent kernel's launch (row-block
                         // tickets, ln_rowblock_tail).  Measured on C1: the tail costs 18 us per
                         // launch against 8.9 us for the LayerNorm launch it replaces, and the code alone, switched off at
                         // run time, slows every persistent launch (+0.38 ms per step) -- so it is compiled out and ln_out
                         // takes a second launch
#endif
#ifndef IMT_WS_RUNAHEAD
#define IMT_WS_RUNAHEAD 0
#endif
#ifndef IMT_GROUP_RUNAHEAD
#define IMT_GROUP_RUNAHEAD 1
#endif
#ifndef IMT_WS_ABLATE
#define IMT_WS_ABLATE 0
#endif
#ifndef IMT_WS_AHEAD
#define IMT_WS_AHEAD true
#endif
#ifndef IMT_WS_ROTATE
#define IMT_WS_ROTATE 0
#endif
#ifndef IMT_GROUP_NST
#define IMT_GROUP_NST 4
#endif


namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int TILE_BYTES = 16384;            // one operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + B

template <typename T, bool KCONTIG> struct TileGeom {
  static constexpr int EPC = 16 / sizeof(T);                  // elements per 16-B chunk
  static constexpr int BK = 128 / sizeof(T);                  // K elements per tile
  static constexpr int RB = KCONTIG ? 128 : 128 * sizeof(T);  // LDS row bytes
  static constexpr int CPR = RB / 16;                         // chunks per row
};

// ------------------------------------------------------------------------------------------------ register staging
template <typename T, bool KCONTIG> struct Stage : TileGeom<T, KCONTIG> {
  using G = TileGeom<T, KCONTIG>;
  u32x4 r[4];
  // KCONTIG: operand[rows = M or N][K], tile rows = 128 operand rows, row chunk c covers k0 + c*EPC
  // else   : operand[K][cols = M or N], tile rows = BK k-rows, row chunk c covers col0 + c*EPC
  IMT_DEVICE void load(const T* __restrict__ base, int64_t ld, int row0, int nrows, int k0, int K) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * NTHREADS + tid;
      const int tr = q / G::CPR, c = q % G::CPR;
      int64_t grow, gcol;
      bool ok;
      if (KCONTIG) { grow = row0 + tr; gcol = k0 + c * G::EPC; ok = (grow < nrows) && (gcol < K); }
      else         { grow = k0 + tr;   gcol = row0 + c * G::EPC; ok = (grow < K) && (gcol < nrows); }
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x4*>(base + grow * ld + gcol);
      r[i] = v;
    }
  }
  IMT_DEVICE void store(char* tile) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * NTHREADS + tid;
      const int tr = q / G::CPR, c = q % G::CPR;
      *reinterpret_cast<u32x4*>(tile + tile_off<G::RB>(tr, c)) = r[i];
    }
  }
};

// ------------------------------------------------------------------------------------------------ LDS-DMA staging
// One wave instruction writes 64 lanes x 16 B = 1 KiB of LDS linearly ("piece"); a 16-KiB tile = 16 pieces, wave w
// issues pieces w, w+4, w+8, w+12.  Lane l of piece p lands on physical chunk q = 64p + l = (row tr, chunk pc); it
// must therefore FETCH logical chunk c = pc ^ swz(tr) of that row (the XOR swizzle is an involution).
template <typename T, bool KCONTIG> struct Dma : TileGeom<T, KCONTIG> {
  using G = TileGeom<T, KCONTIG>;
  __amdgpu_buffer_rsrc_t rsrc;
  int voff[4];    // byte offset of this lane's chunk for k0 = 0
  int kadv;       // byte advance per K tile
  int wv;         // which quarter of the tile's 16 pieces this wave issues
  IMT_DEVICE void init(const T* base, int64_t ld, int64_t valid_bytes, int row0, int kbeg, int wave = -1) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (int)valid_bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    if (wave < 0) wave = threadIdx.x >> 6;
    wv = wave;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = 64 * (4 * i + wave) + lane;
      const int tr = q / G::CPR, pc = q % G::CPR;
      const int c = pc ^ swz<G::RB>(tr);
      if (KCONTIG) voff[i] = (int)((((int64_t)(row0 + tr)) * ld + kbeg + c * G::EPC) * (int64_t)sizeof(T));
      else         voff[i] = (int)((((int64_t)(kbeg + tr)) * ld + row0 + c * G::EPC) * (int64_t)sizeof(T));
    }
    kadv = KCONTIG ? 128 : (int)(G::BK * ld * (int64_t)sizeof(T));
  }
  IMT_DEVICE void issue(char* tile, int t) const {
    const int wave = __builtin_amdgcn_readfirstlane(wv);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(tile + (4 * i + wave) * 1024), 16,
                                               voff[i] + t * kadv, 0, 0, 0);
    }
  }
};

struct EpiParams {
  void* C; int64_t ldc; int c_f32; int accumulate;
  const void* bias; const void* resid; int64_t ldr;
  void* aux; int64_t ldaux; int aux_mode;
  int atomic;
  float alpha; const float* alpha_dev;
  float inv_keep; uint32_t drop_thresh; uint64_t seed;
  float* a_colsum;  // TN only: a_colsum[m] += alpha * sum_k A[k][m]   (bias gradient fused into the dW GEMM)
  int dbg;          // tuning only (tools/gemm_shapes.py): bit0 skip the tile products, bit1 skip the in-loop DMA
  unsigned long long* trace;  // tuning only (IMT_TRACE=gemm_ws | gemm_xl): per-workgroup phase time stamps
  int64_t slab_elems;         // split-K slab mode: C of K split y = C + y * slab_elems (fp32 slabs)
  int splits = 1;             // persistent kernel, split-K slab mode: K ranges per output tile (imt_gemm_args.splitk_ws)
  // in-launch LayerNorm of finished row blocks (persistent kernel, one tile per workgroup; ln_rowblock_tail)
  const void* ln_gamma = nullptr; const void* ln_beta = nullptr; void* ln_out = nullptr; int64_t ld_ln = 0;
  float* ln_mean = nullptr; float* ln_rstd = nullptr; int* ln_tickets = nullptr; float ln_eps = 0.f;
};

// ------------------------------------------------------------------------------------------------ tile product
template <typename T, int LAYOUT>
IMT_DEVICE void load_frags(typename Frag<T>::type (&fa)[4], typename Frag<T>::type (&fb)[4], const char* ta, const char* tb, int wm,
                           int wn, int s) {
  constexpr bool A_KC = (LAYOUT != IMT_TN), B_KC = (LAYOUT == IMT_NT);
  typedef TileGeom<T, A_KC> GA;
  typedef TileGeom<T, B_KC> GB;
  constexpr int KSTEP = Frag<T>::KSTEP;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (A_KC) fa[i] = lds_frag_kcontig<T, GA::RB>(ta, wm + 16 * i, 4 * s);
    else      fa[i] = KStrided<T, GA::RB>::load(ta, s * KSTEP, wm + 16 * i);
    if (B_KC) fb[i] = lds_frag_kcontig<T, GB::RB>(tb, wn + 16 * i, 4 * s);
    else      fb[i] = KStrided<T, GB::RB>::load(tb, s * KSTEP, wn + 16 * i);
  }
}

// the same fragments in the order the i-major MFMA loop consumes them (acc[0][0..3] first: a[0], b[0..3], then a[1..3]): with
// the reads of the next k-step issued one per MFMA, in-order LDS returns then never make an MFMA wait for a recent read
template <typename T, int LAYOUT>
IMT_DEVICE void load_frags_in_use_order(typename Frag<T>::type (&fa)[4], typename Frag<T>::type (&fb)[4], const char* ta,
                                        const char* tb, int wm, int wn, int s) {
  constexpr bool A_KC = (LAYOUT != IMT_TN), B_KC = (LAYOUT == IMT_NT);
  typedef TileGeom<T, A_KC> GA;
  typedef TileGeom<T, B_KC> GB;
  constexpr int KSTEP = Frag<T>::KSTEP;
  auto la = [&](int i) {
    if (A_KC) fa[i] = lds_frag_kcontig<T, GA::RB>(ta, wm + 16 * i, 4 * s);
    else      fa[i] = KStrided<T, GA::RB>::load(ta, s * KSTEP, wm + 16 * i);
  };
  auto lb = [&](int i) {
    if (B_KC) fb[i] = lds_frag_kcontig<T, GB::RB>(tb, wn + 16 * i, 4 * s);
    else      fb[i] = KStrided<T, GB::RB>::load(tb, s * KSTEP, wn + 16 * i);
  };
  lb(0); la(0); lb(1); lb(2); lb(3); la(1); la(2); la(3);
}

// One k-step of the software-pipelined tile product (wave-specialised kernels): the 16 fragment-pair products of the
// fragments in (fa, fb), with the LDS reads of the NEXT k-step -- already written in the source right before this call --
// issued one by one in the shadows of these MFMAs.  NR = LDS read instructions of a k-step for this type / layout.
template <typename T, int LAYOUT> struct StepShape {
  static constexpr bool A_KC = (LAYOUT != IMT_TN), B_KC = (LAYOUT == IMT_NT);
  static constexpr int KS_READS = sizeof(T) == 2 ? 2 : 4;             // K-strided fragment: 2 ds_read_b64_tr_b16 / 4 ds_read_b32
  static constexpr int NR = 4 * (A_KC ? 1 : KS_READS) + 4 * (B_KC ? 1 : KS_READS);
  static constexpr int NM = 16 * (sizeof(T) == 2 ? 1 : 4);            // MFMA instructions of a k-step
  static constexpr int PER = NM / NR > 0 ? NM / NR : 1;               // MFMAs per read slot
};
template <typename T, int LAYOUT, int EXTRA_MFMA = 0>
IMT_DEVICE void mma_step_interleaved(f32x4 (&acc)[4][4], const typename Frag<T>::type (&fa)[4], const typename Frag<T>::type (&fb)[4]) {
  typedef StepShape<T, LAYOUT> S;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) if (!(IMT_WS_ABLATE & 1)) mma16(acc[i][j], fb[j], fa[i]);  // C^T tile: rows <- n, cols <- m
#pragma unroll
  for (int k = 0; k < S::NR; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, S::PER, 0);  // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one DS read
  }
}

// AHEAD = false: fragments of a k-step are read right before its MFMAs (lowest register use; the compiler then tends to
// reuse one register quad for consecutive fragments with a full lgkmcnt(0) between them -- 8 exposed LDS round trips per
// tile, fine where other workgroups of the CU fill the gaps).  AHEAD = true (the wave-specialised kernels: ONE multiplying
// wave per SIMD, nobody else to hide an LDS round trip): every fragment read of the tile is requested before the first
// MFMA (64 VGPRs of fragments for bf16), the scheduling fence keeps the compiler from sinking them back, and the MFMAs of
// step s wait only for the reads of steps <= s (LDS returns in order).
template <typename T, int LAYOUT, bool AHEAD = false>
IMT_DEVICE void compute_tile(f32x4 (&acc)[4][4], const char* ta, const char* tb, int wm, int wn) {
  typedef typename Frag<T>::type frag_t;
  constexpr int NS = TileGeom<T, true>::BK / Frag<T>::KSTEP;  // k-steps per tile (2)
  if (AHEAD) {
    frag_t fa[NS][4], fb[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) load_frags<T, LAYOUT>(fa[s], fb[s], ta, tb, wm, wn, s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma16(acc[i][j], fb[s][j], fa[s][i]);  // C^T tile: rows <- n, cols <- m
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    frag_t fa[4], fb[4];
    load_frags<T, LAYOUT>(fa, fb, ta, tb, wm, wn, s);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) mma16(acc[i][j], fb[j], fa[i]);  // C^T tile: rows <- n, cols <- m
  }
}

// column sums of the K-strided A tile [BK][128] sitting in LDS (TN): thread t owns chunk t % CPR of rows t/CPR + ...
template <typename T> struct ColSum {
  typedef TileGeom<T, false> G;
  static constexpr int EPC = G::EPC;
  float s[EPC];
  IMT_DEVICE void clear() {
#pragma unroll
    for (int e = 0; e < EPC; ++e) s[e] = 0.f;
  }
  // tid: index among the 256 threads that share one 128-column tile (the 256 x 256 kernel has two such groups)
  IMT_DEVICE void add_tile(const char* ta, int tid = threadIdx.x) {
    constexpr int RSTEP = NTHREADS / G::CPR;  // rows covered per pass
    const int c = tid % G::CPR, r0 = tid / G::CPR;
#pragma unroll
    for (int i = 0; i < G::BK / RSTEP; ++i) {
      const int row = r0 + i * RSTEP;
      const typename Frag<T>::type v = *reinterpret_cast<const typename Frag<T>::type*>(ta + tile_off<G::RB>(row, c));
#pragma unroll
      for (int e = 0; e < EPC; ++e) s[e] += (float)v[e];
    }
  }
  // reduce over the threads that share a chunk column, then one atomic per column
  IMT_DEVICE void flush(char* smem, float* out, int m0, int M, float alpha, bool active = true, int tid = threadIdx.x) {
    constexpr int RSTEP = NTHREADS / G::CPR;
    float* red = reinterpret_cast<float*>(smem);  // [RSTEP][128]
    const int c = tid % G::CPR, r0 = tid / G::CPR;
    __syncthreads();
    if (active) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) red[r0 * 128 + c * EPC + e] = s[e];
    }
    __syncthreads();
    if (active && tid < 128) {
      float t = 0.f;
      for (int j = 0; j < RSTEP; ++j) t += red[j * 128 + tid];
      if (m0 + tid < M) atomicAdd(out + m0 + tid, t * alpha);
    }
  }
};

// ------------------------------------------------------------------------------------------------ epilogue
// The accumulators are C^T fragments (lane (lr,lg) of wave (wm,wn) owns C[m = wm+16i+lr][n = wn+16j+4lg..+3]): stored
// directly they would touch 16 rows x 32 bytes per instruction.  Instead the block restages its tile through LDS in
// two passes of 64 rows (fp32, 32 KiB, 16-byte groups XOR-swizzled by row) so that EVERY global access of the epilogue
// (bias, residual, GELU aux, C) is a 16-B-per-lane (8 B for bf16 stores) access with 32 consecutive lanes per row.
IMT_DEVICE void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
IMT_DEVICE int epi_off(int row, int c4) { return row * 512 + ((c4 ^ (row & 31)) << 4); }

// NTHR threads share the group loop (256 = the MFMA waves alone; 512 = the persistent kernel's LAST tile, where its idle
// producer waves help: HAS_ACC = false for them -- they hold no accumulators and only join from the first barrier on).
// WROWS = 128 (the 256 x 256 kernel): an owning wave holds all 128 rows of the tile's 64-column strip, acc[8][4], and
// restages rows 64*pass .. +63 in each pass instead of sitting one out.
//
// CODE SIZE is a first-order cost here: the group loop is unrolled (4-8 bodies per pass), and a body that branches over
// every option at run time (GELU, GELU', dropout, residual, fp32 / accumulate, ragged-N scalar tail) is > 1000
// instructions of which a launch executes a few dozen -- the wave then takes an instruction-cache miss at every
// skipped block (measured with IMT_GEMM_TRACE on the 256-tile kernel: 12-18 us per tile for what is 3 us of work).
// So the epilogue KIND is resolved once per tile (uniform) into a straight-line template instance for the kinds the
// train step uses on full tiles; everything else (edge tiles, split-K atomics, rare combinations) runs ONE rolled,
// fully general body.
enum { EM_PLAIN = 0, EM_RESID, EM_DROP_RESID, EM_GELU, EM_DGELU, EM_F32, EM_F32_ACC, EM_ACC, EM_GENERIC };

IMT_DEVICE int epi_kind(const EpiParams& ep) {
  const bool extras = ep.resid || ep.drop_thresh;
  if (ep.atomic) return EM_GENERIC;
  if (ep.c_f32) return (ep.aux_mode != IMT_AUX_NONE || extras) ? EM_GENERIC : (ep.accumulate ? EM_F32_ACC : EM_F32);
  if (ep.accumulate) return (ep.aux_mode != IMT_AUX_NONE || extras) ? EM_GENERIC : EM_ACC;
  if (ep.aux_mode == IMT_AUX_GELU_FWD) return extras ? EM_GENERIC : EM_GELU;
  if (ep.aux_mode == IMT_AUX_DGELU) return extras ? EM_GENERIC : EM_DGELU;
  if (ep.drop_thresh) return ep.resid ? EM_DROP_RESID : EM_GENERIC;
  return ep.resid ? EM_RESID : EM_PLAIN;
}

// restage this wave's part of rows 64*pass .. +63 (fp32, swizzled); between the two LDS-only barriers.  __syncthreads()
// would also drain vmcnt, i.e. make every pass wait for the previous pass's global stores to be acknowledged.
template <bool HAS_ACC, int WROWS>
IMT_DEVICE void epi_restage(const f32x4 (*acc)[4], char* smem, int pass, int wm, int wn, float alpha) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
  lds_barrier();  // previous users of smem (K loop / previous pass) are done
  if (HAS_ACC && (WROWS == 128 || wm == 64 * pass)) {
    const int ib = (WROWS == 128) ? 4 * pass : 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(smem + epi_off(16 * i + lr, (wn + 16 * j + 4 * lg) >> 2)) = acc[ib + i][j] * alpha;
  }
  lds_barrier();
}

// full tile, kind known at compile time: no bounds checks, no branches; the per-element operands of a pass (GELU' input,
// residual, old C) are requested before the tile is restaged and stay in flight across the two barriers
template <typename T, int KIND, int NTHR, bool HAS_ACC, int WROWS, bool RAGGED_M>
IMT_DEVICE void epilogue_fast(const f32x4 (*acc)[4], char* smem, int m0, int n0, int wm, int wn, int M, int N, const EpiParams& ep, float alpha) {
  typedef typename Vec4<T>::type raw_t;
  constexpr int GPT = 2048 / NTHR;  // 4-column groups per thread per 64-row pass
  constexpr bool C32 = (KIND == EM_F32 || KIND == EM_F32_ACC);
  const int c4 = threadIdx.x & 31, n = n0 + 4 * c4;
  const T* bias = reinterpret_cast<const T*>(ep.bias);
  const T* resid = reinterpret_cast<const T*>(ep.resid);
  T* aux = reinterpret_cast<T*>(ep.aux);
  const f32x4 bv = bias ? Vec4<T>::load(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    raw_t pr[GPT];   // GELU' input / residual / old C (same type as the operands)
    f32x4 pc[KIND == EM_F32_ACC ? GPT : 1];
    const int64_t mrow = m0 + 64 * pass + ((int)threadIdx.x >> 5);  // + gq * (NTHR / 32)
#pragma unroll
    for (int gq = 0; gq < GPT; ++gq) {
      // ragged M (8128 target rows): loads of the rows past M read row M-1 instead (no control flow around a load: a
      // branch makes the compiler drain vmcnt at every join), only the stores below are predicated
      const int64_t m = RAGGED_M ? min((int64_t)(mrow + gq * (NTHR / 32)), (int64_t)M - 1) : mrow + gq * (NTHR / 32);
      if (KIND == EM_DGELU) pr[gq] = Vec4<T>::load_raw(aux + m * ep.ldaux + n);
      if (KIND == EM_RESID || KIND == EM_DROP_RESID) pr[gq] = Vec4<T>::load_raw(resid + m * ep.ldr + n);
      if (KIND == EM_ACC) pr[gq] = Vec4<T>::load_raw(reinterpret_cast<const T*>(ep.C) + m * ep.ldc + n);
      if (KIND == EM_F32_ACC) pc[gq] = Vec4<float>::load(reinterpret_cast<const float*>(ep.C) + m * ep.ldc + n);
    }
    epi_restage<HAS_ACC, WROWS>(acc, smem, pass, wm, wn, alpha);
#pragma unroll
    for (int gq = 0; gq < GPT; ++gq) {
      const int row = ((int)threadIdx.x >> 5) + gq * (NTHR / 32);
      const int64_t m = mrow + gq * (NTHR / 32);
      const bool live = !RAGGED_M || m < M;
      f32x4 v = *reinterpret_cast<const f32x4*>(smem + epi_off(row, c4)) + bv;
      if (KIND == EM_GELU) {
        if (live) Vec4<T>::store(aux + m * ep.ldaux + n, v);
        v = gelu_erf4(v);
      }
      if (KIND == EM_DGELU) {
        const f32x4 z = Vec4<T>::cvt(pr[gq]);
        v *= gelu_erf_grad4(z);
      }
      if (KIND == EM_DROP_RESID) {
        if ((N & 3) == 0) {  // n % 4 == 0: the lane's four elements are one block of the dropout generator
          dropout_apply4(v, ep.seed, (uint64_t)m * (uint64_t)N + (uint64_t)n, ep.drop_thresh, ep.inv_keep);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = dropout_keep(ep.seed, (uint64_t)m * (uint64_t)N + (uint64_t)(n + e), ep.drop_thresh) ? v[e] * ep.inv_keep : 0.f;
        }
      }
      if (KIND == EM_RESID || KIND == EM_DROP_RESID || KIND == EM_ACC) v += Vec4<T>::cvt(pr[gq]);
      if (KIND == EM_F32_ACC) v += pc[gq];
      if (live) {
        // (non-temporal stores here, to leave less dirty data for the end-of-kernel write-back: no measurable change)
        if (IMT_LN_TICKET && ep.ln_out) {  // read by another workgroup of this launch: write-through (uniform branch)
          if (C32) Vec4<float>::store_wt(reinterpret_cast<float*>(ep.C) + m * ep.ldc + n, v);
          else     Vec4<T>::store_wt(reinterpret_cast<T*>(ep.C) + m * ep.ldc + n, v);
        } else {
          if (C32) Vec4<float>::store(reinterpret_cast<float*>(ep.C) + m * ep.ldc + n, v);
          else     Vec4<T>::store(reinterpret_cast<T*>(ep.C) + m * ep.ldc + n, v);
        }
      }
    }
  }
}

// any tile, any option: one rolled body
template <typename T, int NTHR, bool HAS_ACC, int WROWS>
IMT_DEVICE void epilogue_general(const f32x4 (*acc)[4], char* smem, int m0, int n0, int wm, int wn, int M, int N, const EpiParams& ep,
                                 float alpha) {
  const T* bias = reinterpret_cast<const T*>(ep.bias);
  const T* resid = reinterpret_cast<const T*>(ep.resid);
  T* aux = reinterpret_cast<T*>(ep.aux);
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0) epi_restage<HAS_ACC, WROWS>(acc, smem, 0, wm, wn, alpha);
    else           epi_restage<HAS_ACC, WROWS>(acc, smem, 1, wm, wn, alpha);
#pragma unroll 1
    for (int idx = threadIdx.x; idx < 2048; idx += NTHR) {
      const int row = idx >> 5, c4 = idx & 31;
      const int m = m0 + 64 * pass + row, n = n0 + 4 * c4;
      if (m >= M || n >= N) continue;
      const f32x4 v4 = *reinterpret_cast<const f32x4*>(smem + epi_off(row, c4));
      if (!ep.atomic && n + 3 < N) {  // whole 4-column group inside the matrix: vector accesses
        f32x4 v = v4;
        if (bias) v += Vec4<T>::load(bias + n);
        if (ep.aux_mode == IMT_AUX_GELU_FWD) {
          Vec4<T>::store(aux + (int64_t)m * ep.ldaux + n, v);
          v = gelu_erf4(v);
        } else if (ep.aux_mode == IMT_AUX_DGELU) {
          const f32x4 z = Vec4<T>::load(aux + (int64_t)m * ep.ldaux + n);
          v *= gelu_erf_grad4(z);
        }
        if (ep.drop_thresh) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = dropout_keep(ep.seed, (uint64_t)m * (uint64_t)N + (uint64_t)(n + e), ep.drop_thresh) ? v[e] * ep.inv_keep : 0.f;
        }
        if (resid) v += Vec4<T>::load(resid + (int64_t)m * ep.ldr + n);
        if (ep.c_f32) {
          float* c = reinterpret_cast<float*>(ep.C) + (int64_t)m * ep.ldc + n;
          if (ep.accumulate) v += Vec4<float>::load(c);
          if (IMT_LN_TICKET && ep.ln_out) Vec4<float>::store_wt(c, v); else Vec4<float>::store(c, v);
        } else {
          T* c = reinterpret_cast<T*>(ep.C) + (int64_t)m * ep.ldc + n;
          if (ep.accumulate) v += Vec4<T>::load(c);
          if (IMT_LN_TICKET && ep.ln_out) Vec4<T>::store_wt(c, v); else Vec4<T>::store(c, v);
        }
        continue;
      }
#pragma unroll 1
      for (int e = 0; e < 4; ++e) {
        if (n + e >= N) break;
        float x = e == 0 ? v4[0] : e == 1 ? v4[1] : e == 2 ? v4[2] : v4[3];
        if (ep.atomic) { atomicAdd(reinterpret_cast<float*>(ep.C) + (int64_t)m * ep.ldc + n + e, x); continue; }
        if (bias) x += to_f32<T>(bias[n + e]);
        if (ep.aux_mode == IMT_AUX_GELU_FWD) {
          aux[(int64_t)m * ep.ldaux + n + e] = from_f32<T>(x);
          x = gelu_erf(x);
        } else if (ep.aux_mode == IMT_AUX_DGELU) {
          x *= gelu_erf_grad(to_f32<T>(aux[(int64_t)m * ep.ldaux + n + e]));
        }
        if (ep.drop_thresh)
          x = dropout_keep(ep.seed, (uint64_t)m * (uint64_t)N + (uint64_t)(n + e), ep.drop_thresh) ? x * ep.inv_keep : 0.f;
        if (resid) x += to_f32<T>(resid[(int64_t)m * ep.ldr + n + e]);
        if (ep.c_f32) {
          float* c = reinterpret_cast<float*>(ep.C) + (int64_t)m * ep.ldc + n + e;
          if (ep.accumulate) x += *c;
          *c = x;
        } else {
          T* c = reinterpret_cast<T*>(ep.C) + (int64_t)m * ep.ldc + n + e;
          if (ep.accumulate) x += to_f32<T>(*c);
          *c = from_f32<T>(x);
        }
      }
    }
  }
}

template <typename T, int NTHR, bool HAS_ACC, int WROWS, bool RAGGED_M>
IMT_DEVICE void epilogue_fast_kinds(int kind, const f32x4 (*acc)[4], char* smem, int m0, int n0, int wm, int wn, int M, int N,
                                    const EpiParams& ep, float alpha) {
  switch (kind) {
    case EM_PLAIN:      epilogue_fast<T, EM_PLAIN, NTHR, HAS_ACC, WROWS, RAGGED_M>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha); break;
    case EM_RESID:      epilogue_fast<T, EM_RESID, NTHR, HAS_ACC, WROWS, RAGGED_M>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha); break;
    case EM_DROP_RESID: epilogue_fast<T, EM_DROP_RESID, NTHR, HAS_ACC, WROWS, RAGGED_M>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha); break;
    case EM_GELU:       epilogue_fast<T, EM_GELU, NTHR, HAS_ACC, WROWS, RAGGED_M>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha); break;
    case EM_DGELU:      epilogue_fast<T, EM_DGELU, NTHR, HAS_ACC, WROWS, RAGGED_M>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha); break;
    case EM_F32:        epilogue_fast<T, EM_F32, NTHR, HAS_ACC, WROWS, RAGGED_M>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha); break;
    case EM_F32_ACC:    epilogue_fast<T, EM_F32_ACC, NTHR, HAS_ACC, WROWS, RAGGED_M>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha); break;
    default:            epilogue_fast<T, EM_ACC, NTHR, HAS_ACC, WROWS, RAGGED_M>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha); break;
  }
}

template <typename T, int NTHR = NTHREADS, bool HAS_ACC = true, int WROWS = 64>
IMT_DEVICE void epilogue(const f32x4 (*acc)[4], char* smem, int m0, int n0, int wm, int wn, int M, int N, const EpiParams& ep,
                         float alpha) {
  const int kind = (n0 + BN <= N) ? epi_kind(ep) : EM_GENERIC;  // uniform over the workgroup
  if (kind == EM_GENERIC || ((ep.dbg & 16) && m0 + BM > M)) epilogue_general<T, NTHR, HAS_ACC, WROWS>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha);
  else if (m0 + BM <= M)  epilogue_fast_kinds<T, NTHR, HAS_ACC, WROWS, false>(kind, acc, smem, m0, n0, wm, wn, M, N, ep, alpha);
  else                    epilogue_fast_kinds<T, NTHR, HAS_ACC, WROWS, true>(kind, acc, smem, m0, n0, wm, wn, M, N, ep, alpha);
}

// XCD-aware block order (T1): consecutive ids on one XCD walk neighbouring tiles (bijective for any grid size).
IMT_DEVICE void tile_origin(int M, int N, int& m0, int& n0) {
  const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM;
  const int nwg = nbx * nby;
  const int bid = imt_xcd_block(blockIdx.x, nwg);
  m0 = (bid / nbx) * BM;
  n0 = (bid % nbx) * BN;
}

// ------------------------------------------------------------------------------------------------ general kernel
template <typename T, int LAYOUT>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                                        int64_t ldb, int M, int N, int K, int k_per_split, EpiParams ep) {
  constexpr bool A_KC = (LAYOUT != IMT_TN), B_KC = (LAYOUT == IMT_NT);
  typedef Stage<T, A_KC> SA;
  typedef Stage<T, B_KC> SB;
  constexpr int BK = SA::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // buffer b: A tile at b*32 KiB, B tile 16 KiB later

  int m0, n0;
  tile_origin(M, N, m0, n0);
  const int kbeg = blockIdx.y * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  const int wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  ColSum<T> cs;
  cs.clear();
  const bool do_colsum = (LAYOUT == IMT_TN) && ep.a_colsum && n0 == 0;

  SA sa; SB sb;
  const int nt = (kend - kbeg + BK - 1) / BK;
  if (nt > 0) {
    sa.load(A, lda, m0, M, kbeg, kend);
    sb.load(B, ldb, n0, N, kbeg, kend);
    sa.store(smem);
    sb.store(smem + TILE_BYTES);
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      sa.load(A, lda, m0, M, kbeg + (t + 1) * BK, kend);
      sb.load(B, ldb, n0, N, kbeg + (t + 1) * BK, kend);
    }
    const char* ta = smem + cur * STAGE_BYTES;
    compute_tile<T, LAYOUT>(acc, ta, ta + TILE_BYTES, wm, wn);
    if (LAYOUT == IMT_TN && do_colsum) cs.add_tile(ta);
    if (t + 1 < nt) {
      sa.store(smem + (cur ^ 1) * STAGE_BYTES);
      sb.store(smem + (cur ^ 1) * STAGE_BYTES + TILE_BYTES);
    }
    __syncthreads();
  }
  const float alpha = ep.alpha_dev ? ep.alpha * ep.alpha_dev[0] : ep.alpha;
  epilogue<T>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha);
  if (LAYOUT == IMT_TN && do_colsum) cs.flush(smem, ep.a_colsum, m0, M, alpha);
}

// ------------------------------------------------------------------------------------------------ high-occupancy kernel
// Single 32-KiB LDS buffer + register prefetch: two barriers per K tile, but three workgroups (12 waves) per CU,
// i.e. three tiles of loads in flight per CU and other blocks' MFMAs to fill every barrier / latency bubble.
template <typename T, int LAYOUT>
__global__ __launch_bounds__(NTHREADS, 3) void gemm_sb_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                                              int64_t ldb, int M, int N, int K, int k_per_split, EpiParams ep) {
  constexpr bool A_KC = (LAYOUT != IMT_TN), B_KC = (LAYOUT == IMT_NT);
  typedef Stage<T, A_KC> SA;
  typedef Stage<T, B_KC> SB;
  constexpr int BK = SA::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int m0, n0;
  tile_origin(M, N, m0, n0);
  const int kbeg = blockIdx.y * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  const int wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  ColSum<T> cs;
  cs.clear();
  const bool do_colsum = (LAYOUT == IMT_TN) && ep.a_colsum && n0 == 0;
  SA sa; SB sb;
  const int nt = (kend - kbeg + BK - 1) / BK;
  if (nt > 0) {
    sa.load(A, lda, m0, M, kbeg, kend);
    sb.load(B, ldb, n0, N, kbeg, kend);
  }
  for (int t = 0; t < nt; ++t) {
    sa.store(smem);
    sb.store(smem + TILE_BYTES);
    __syncthreads();
    if (t + 1 < nt) {
      sa.load(A, lda, m0, M, kbeg + (t + 1) * BK, kend);
      sb.load(B, ldb, n0, N, kbeg + (t + 1) * BK, kend);
    }
    compute_tile<T, LAYOUT>(acc, smem, smem + TILE_BYTES, wm, wn);
    if (LAYOUT == IMT_TN && do_colsum) cs.add_tile(smem);
    __syncthreads();
  }
  const float alpha = ep.alpha_dev ? ep.alpha * ep.alpha_dev[0] : ep.alpha;
  epilogue<T>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha);
  if (LAYOUT == IMT_TN && do_colsum) cs.flush(smem, ep.a_colsum, m0, M, alpha);
}

// ------------------------------------------------------------------------------------------------ pipelined kernel
constexpr int NST = 3;  // LDS ring depth (96 KiB): tile t+2 in flight while tile t is multiplied

template <typename T, int LAYOUT, int NSTG>
__global__ __launch_bounds__(NTHREADS) void gemm_pipe_kernel(const T* __restrict__ A, int64_t lda, int64_t a_bytes,
                                                             const T* __restrict__ B, int64_t ldb, int64_t b_bytes, int M, int N,
                                                             int K, int k_per_split, EpiParams ep) {
  constexpr bool A_KC = (LAYOUT != IMT_TN), B_KC = (LAYOUT == IMT_NT);
  constexpr int BK = TileGeom<T, A_KC>::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // ALL LDS in this one array (second-object trap)

  int m0, n0;
  tile_origin(M, N, m0, n0);
  const int kbeg = blockIdx.y * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  const int nt = (kend - kbeg) / BK;  // host guarantees (kend - kbeg) % BK == 0
  const int wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  Dma<T, A_KC> da; Dma<T, B_KC> db;
  da.init(A, lda, a_bytes, m0, kbeg);
  db.init(B, ldb, b_bytes, n0, kbeg);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  ColSum<T> cs;
  cs.clear();
  const bool do_colsum = (LAYOUT == IMT_TN) && ep.a_colsum && n0 == 0;

  // K-order rotation: workgroups that run at the same time start at different K offsets, so they do not all hit the
  // same L2 / memory channels in lock-step (row strides are powers of two).  Summation order differs per block only.
  const int phase = nt > 0 ? (int)((blockIdx.x * 7u + blockIdx.y * 3u) % (unsigned)nt) : 0;
  auto ktile = [&](int t) { const int k = t + phase; return k >= nt ? k - nt : k; };
  // prologue: tiles 0 .. NSTG-2 in flight (8 DMA instructions per wave per tile: 4 for A, 4 for B)
#pragma unroll
  for (int s0 = 0; s0 < NSTG - 1; ++s0)
    if (s0 < nt) { da.issue(smem + s0 * STAGE_BYTES, ktile(s0)); db.issue(smem + s0 * STAGE_BYTES + TILE_BYTES, ktile(s0)); }
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    // tile t has landed once all but the (NSTG-2) newest tiles' DMAs are done; the barrier then (a) publishes every
    // wave's pieces of tile t and (b) proves every wave finished reading the buffer the next DMA overwrites.
    const int newer = min(NSTG - 2, nt - 1 - t);
    if (newer >= 2)      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (newer == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (t + NSTG - 1 < nt && !(ep.dbg & 2)) {
      const int nxt = (cur == 0) ? NSTG - 1 : cur - 1;  // (cur + NSTG - 1) % NSTG
      da.issue(smem + nxt * STAGE_BYTES, ktile(t + NSTG - 1));
      db.issue(smem + nxt * STAGE_BYTES + TILE_BYTES, ktile(t + NSTG - 1));
    }
    const char* ta = smem + cur * STAGE_BYTES;
    if (!(ep.dbg & 1)) compute_tile<T, LAYOUT>(acc, ta, ta + TILE_BYTES, wm, wn);
    if (LAYOUT == IMT_TN && do_colsum) cs.add_tile(ta);
    cur = (cur + 1 == NSTG) ? 0 : cur + 1;
  }
  const float alpha = ep.alpha_dev ? ep.alpha * ep.alpha_dev[0] : ep.alpha;
  epilogue<T>(acc, smem, m0, n0, wm, wn, M, N, ep, alpha);
  if (LAYOUT == IMT_TN && do_colsum) cs.flush(smem, ep.a_colsum, m0, M, alpha);
}


// ------------------------------------------------------------------------------------------------ in-launch LayerNorm
// LayerNorm(dropout(dense(x)) + input) without a second launch (src/bert_seq2seq.py:84-90,139-143).  A row needs all N
// columns, i.e. the nbx column tiles of its 128-row block, which different workgroups compute.  No workgroup waits for
// another: every workgroup stores its C tile write-through (sc1), drains its stores, and one lane takes a ticket of the row
// block with an agent-scope atomic add; the workgroup whose ticket is the last one (nbx - 1) knows that all tiles of the
// block are in memory and normalises the block's rows, reading C around its L1 / L2 (sc1 loads: the other tiles were
// written by other CUs).  This is the "last arriver" hand-off of MI355X_MICROARCH.md (section Workgroup dispatch ...:
// every store of the handed-off bytes sc1 and drained by its wave before the workgroup's one atomic add; every load of them
// an sc1 buffer load to registers issued after the add has returned and a workgroup barrier).  The ticket returns to zero
// for the next launch.  All WS_THREADS threads of the workgroup call this (barriers inside).
// the rows of one finished block: each wave takes 16 of the 128 rows, eight at a time with all their loads in flight
// (a row is one dependent load -> reduce -> store chain: taken one by one, sixteen memory round trips per wave)
template <typename T, int NCH>
IMT_DEVICE void ln_finished_rows(const EpiParams& ep, int m0, int rows, int M, int N) {
  constexpr int EPL = 16 / sizeof(T);  // elements per lane and chunk (one 16-byte load)
  constexpr int CW = 64 * EPL;         // columns per chunk
  constexpr int RB = 8;                // rows in flight per wave
  typedef typename Frag<T>::type vec_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // one descriptor over the whole C view; offsets stay below 2^31 (host-checked)
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(ep.C, 0, (int)(((int64_t)(M - 1) * ep.ldc + N) * sizeof(T)), 0x00020000);
  float g[NCH][EPL], b[NCH][EPL];
  int col[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    col[i] = min(lane * EPL + i * CW, N - EPL);  // clamped: loads unconditional, masked at use
    const vec_t gv = *reinterpret_cast<const vec_t*>(reinterpret_cast<const T*>(ep.ln_gamma) + col[i]);
    const vec_t bv = *reinterpret_cast<const vec_t*>(reinterpret_cast<const T*>(ep.ln_beta) + col[i]);
#pragma unroll
    for (int e = 0; e < EPL; ++e) { g[i][e] = (float)gv[e]; b[i][e] = (float)bv[e]; }
  }
  for (int r0 = wave * 16; r0 < rows; r0 += 16 * 8) {  // (128 rows, 8 waves: one trip)
#pragma unroll
    for (int h = 0; h < 16; h += RB) {
      u32x4 raw[RB][NCH];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int64_t m = m0 + min(r0 + h + k, rows - 1);  // clamped row: read, not used
#pragma unroll
        for (int i = 0; i < NCH; ++i)
          raw[k][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rc, (int)((m * ep.ldc + col[i]) * sizeof(T)), 0, 16 /* sc1 */));
      }
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int r = r0 + h + k;
        const int64_t m = m0 + r;
        float v[NCH][EPL];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const vec_t x = __builtin_bit_cast(vec_t, raw[k][i]);
          const bool live = lane * EPL + i * CW < N;
#pragma unroll
          for (int e = 0; e < EPL; ++e) { v[i][e] = (float)x[e]; if (live) s += v[i][e]; }
        }
        const float mean = wave_sum(s) / (float)N;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
          if (lane * EPL + i * CW < N) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) { const float t = v[i][e] - mean; q += t * t; }
          }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)N + ep.ln_eps);
        if (r < rows) {
          if (lane == 0) {
            if (ep.ln_mean) ep.ln_mean[m] = mean;
            if (ep.ln_rstd) ep.ln_rstd[m] = rstd;
          }
          T* yr = reinterpret_cast<T*>(ep.ln_out) + m * ep.ld_ln;
#pragma unroll
          for (int i = 0; i < NCH; ++i) {
            const int c = lane * EPL + i * CW;
            if (c < N) {
              vec_t o;
#pragma unroll
              for (int e = 0; e < EPL; ++e) o[e] = from_f32<T>((v[i][e] - mean) * rstd * g[i][e] + b[i][e]);
              *reinterpret_cast<vec_t*>(yr + c) = o;
            }
          }
        }
      }
    }
  }
}

// (not inlined: inlined into the persistent kernel its registers pushed the kernel past the 256 a wave gets at two waves per
// SIMD and the main loop spilled)
template <typename T>
__device__ __attribute__((noinline)) void ln_rowblock_tail(const EpiParams& ep, char* lds, int m0, int row_block, int nbx, int M, int N) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's C stores have completed (written through)
  __syncthreads();
  int* flag = reinterpret_cast<int*>(lds);
  if (threadIdx.x == 0) *flag = __hip_atomic_fetch_add(ep.ln_tickets + row_block, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*flag != nbx - 1) return;  // uniform
  if (threadIdx.x == 0) __hip_atomic_store(ep.ln_tickets + row_block, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int rows = min(BM, M - m0);
  const int nch = (N + 1024 / (int)sizeof(T) - 1) / (1024 / (int)sizeof(T));  // 16-byte loads: 512 bf16 / 256 fp32 columns per chunk
  if (nch <= 1)      ln_finished_rows<T, 1>(ep, m0, rows, M, N);
  else if (nch == 2) ln_finished_rows<T, 2>(ep, m0, rows, M, N);
  else               ln_finished_rows<T, 4>(ep, m0, rows, M, N);  // N <= 1024 (host-checked)
}

// ------------------------------------------------------------------------------------------------ persistent wave-specialised kernel
// The main GEMM loop of the path when K is a whole number of tiles: 512 threads, waves 0-3 multiply (64x64 each),
// waves 4-7 stream the operand tiles by LDS-DMA into a 3-stage ring and run AHEAD ACROSS OUTPUT TILES (one workgroup
// per CU walks tiles b, b+grid, ...), so neither the DMA issue cost nor the first-tile latency nor the epilogue of a
// tile stalls the MFMA waves of the next one.  The epilogue restages through its own 32 KiB (ring 96 KiB + 32 KiB
// = 128 KiB: a communication workgroup of a data-parallel job can share the CU).  Producers keep their own counted vmcnt (only DMA), consumers' epilogue loads/stores
// have theirs; the only coupling is one raw s_barrier per K tile plus the epilogue's four.
constexpr int WS_NST = IMT_WS_NST;   // 96 KiB ring + 32 KiB epilogue staging = 128 KiB: leaves LDS for a co-resident communication workgroup
constexpr int WS_THREADS = 512;
constexpr int WS_LDS = WS_NST * STAGE_BYTES + 32768;

template <typename T, int LAYOUT>
__global__ __launch_bounds__(WS_THREADS) void gemm_ws_kernel(const T* __restrict__ A, int64_t lda, int64_t a_bytes,
                                                             const T* __restrict__ B, int64_t ldb, int64_t b_bytes, int M, int N,
                                                             int K, EpiParams ep) {
  constexpr bool A_KC = (LAYOUT != IMT_TN), B_KC = (LAYOUT == IMT_NT);
  constexpr int BK = TileGeom<T, A_KC>::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* epi = smem + WS_NST * STAGE_BYTES;
  const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM;
  // split-K slab mode (ep.splits > 1): a work item is (output tile, K range); items of one tile are neighbours in the XCD-affine
  // order, so its slabs are written by one XCD.  Every range holds at least one K tile (host-checked).
  const int S = ep.splits > 1 ? ep.splits : 1;
  const int tiles = nbx * nby * S;
  // whole K tiles for NT / NN (host-checked).  TN (weight gradients: K = token count, any value): both operands are K-STRIDED,
  // so the rows of a ragged last K tile lie past the operands' valid bytes and the descriptors' range check fills them with
  // zeros -- they add nothing to the products (or to the fused column sums)
  const int nt_all = (K + BK - 1) / BK;
  const int per = (nt_all + S - 1) / S;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool consumer = wave < 4;
  const int my_tiles = ((int)blockIdx.x < tiles) ? (tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const bool colsum_kernel = (LAYOUT == IMT_TN) && ep.a_colsum;
  // work item i of this workgroup -> output tile lt, K range [kt0, kt0 + ntl) in K tiles, slab ks
  const auto item = [&](int i, int& lt, int& kt0, int& ntl) {
    const int v = imt_xcd_block(blockIdx.x + i * gridDim.x, tiles);
    lt = v / S;
    const int ks = v - lt * S;
    kt0 = ks * per;
    ntl = min(per, nt_all - kt0);
    return ks;
  };

  if (!consumer) {
    // ------------------------------------------------------------ producers: flat stream of (tile, k) steps
    Dma<T, A_KC> da; Dma<T, B_KC> db;
    int total = 0;
    for (int i = 0; i < my_tiles; ++i) { int lt, kt0, ntl; item(i, lt, kt0, ntl); total += ntl; }
    int iq = 0, tq = 0;  // item / k index of the NEXT step to issue
    int rot = 0, nt_q = 0;
    auto issue_next = [&](int slot) {
      if (tq == 0) {
        int lt, kt0;
        item(iq, lt, kt0, nt_q);
        da.init(A, lda, a_bytes, (lt / nbx) * BM, kt0 * BK, wave - 4);
        db.init(B, ldb, b_bytes, (lt % nbx) * BN, kt0 * BK, wave - 4);
        // (experiment) the workgroups that share an operand row block start at different K tiles, so a tile is
        // fetched from the Infinity Cache by one of them and found in the XCD's L2 by the others
        if (IMT_WS_ROTATE && sizeof(T) == 2 && S == 1) rot = ((lt % nbx) * nt_q / nbx + (lt / nbx)) % nt_q;
      }
      int tk = tq + rot;
      if (tk >= nt_q) tk -= nt_q;
      if (!(IMT_WS_ABLATE & 2)) {  // (-DIMT_WS_ABLATE=2: no DMA, the consumers multiply whatever the ring holds; 3: bare loop)
        da.issue(smem + slot * STAGE_BYTES, tk);
        db.issue(smem + slot * STAGE_BYTES + TILE_BYTES, tk);
      }
      if (++tq == nt_q) { tq = 0; ++iq; }
    };
    int issued = 0;
#pragma unroll
    for (int s0 = 0; s0 < WS_NST - 1; ++s0)
      if (issued < total) { issue_next(s0); ++issued; }
    int cur = 0, t = 0;
    const auto wait_newer = [](int newer) {  // all of this wave's DMA pieces done but those of the `newer` youngest steps
      if (newer >= 3)      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (newer == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (IMT_WS_RUNAHEAD && total > 0) {  // run-ahead protocol: barrier q publishes step q + 1; one more barrier in front publishes step 0
      wait_newer(issued - 1);
      asm volatile("s_barrier" ::: "memory");
    }
    int ci = 0, nt_c = 0;  // item being consumed and its K tiles
    if (my_tiles > 0) { int lt, kt0; item(0, lt, kt0, nt_c); }
    for (int q = 0; q < total; ++q) {
      // steps issued after the step this barrier publishes (classic: step q; run-ahead: step q + 1)
      wait_newer(issued - 1 - q - (IMT_WS_RUNAHEAD ? 1 : 0));
      asm volatile("s_barrier" ::: "memory");
      if (issued < total) { issue_next(cur == 0 ? WS_NST - 1 : cur - 1); ++issued; }
      cur = (cur + 1 == WS_NST) ? 0 : cur + 1;
      if (++t == nt_c) {
        t = 0;
        if (q == total - 1) {
          // last tile of this workgroup: nothing left to stream, so the producer waves take half of the epilogue's
          // group loop (with one tile per CU -- every N = 512 GEMM of the step -- the epilogue is fully exposed)
          int lt, kt0, ntl;
          const int ks = item(my_tiles - 1, lt, kt0, ntl);
          const f32x4 none[4][4] = {};
          const float alpha = ep.alpha_dev ? ep.alpha * ep.alpha_dev[0] : ep.alpha;
          EpiParams eps = ep;
          if (S > 1) eps.C = reinterpret_cast<float*>(ep.C) + (int64_t)ks * ep.slab_elems;
          epilogue<T, 512, false>(none, epi, (lt / nbx) * BM, (lt % nbx) * BN, -1, 0, M, N, eps, alpha);
          if (IMT_LN_TICKET && ep.ln_out) ln_rowblock_tail<T>(ep, epi, (lt / nbx) * BM, lt / nbx, nbx, M, N);  // (host: one tile per workgroup)
        } else {
          // the consumers' epilogue: 2 passes x 2 barriers
          asm volatile("s_barrier\n\ts_barrier\n\ts_barrier\n\ts_barrier" ::: "memory");
        }
        if (colsum_kernel) asm volatile("s_barrier\n\ts_barrier" ::: "memory");  // + the fused column sums
        if (++ci < my_tiles) { int lt, kt0; item(ci, lt, kt0, nt_c); }
      }
    }
  } else {
    // ------------------------------------------------------------ consumers
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const float alpha = ep.alpha_dev ? ep.alpha * ep.alpha_dev[0] : ep.alpha;
    int cur = 0;
    IMT_STAMP(ep.trace, 0);
    if (IMT_WS_RUNAHEAD && my_tiles * nt_all > 0) asm volatile("s_barrier" ::: "memory");  // step 0 landed (same condition as the producers' total > 0; run-ahead builds: no split-K)
    for (int i = 0; i < my_tiles; ++i) {
      int lt, kt0, nt;
      const int ks = item(i, lt, kt0, nt);
      const int m0 = (lt / nbx) * BM, n0 = (lt % nbx) * BN;
      f32x4 acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      ColSum<T> cs;
      cs.clear();
      const bool do_colsum = colsum_kernel && n0 == 0;
#if IMT_WS_RUNAHEAD
      // software pipeline over k-steps (see grouped_consumer_loop): barrier q has told us that K tile q + 1 has landed, so
      // its first fragments are read while tile q is still being multiplied.  Not across output tiles: the fragments would
      // be live through the epilogue.
      typename Frag<T>::type f0a[4], f0b[4], f1a[4], f1b[4];
      {
        const char* t0 = smem + cur * STAGE_BYTES;  // first K tile of this output tile: landed (initial barrier / the previous K tile's)
        load_frags_in_use_order<T, LAYOUT>(f0a, f0b, t0, t0 + TILE_BYTES, wm, wn, 0);
      }
      for (int t = 0; t < nt; ++t) {
        if (i == 0 && t == 0) IMT_STAMP(ep.trace, 1);
        const char* ta = smem + cur * STAGE_BYTES;
        load_frags_in_use_order<T, LAYOUT>(f1a, f1b, ta, ta + TILE_BYTES, wm, wn, 1);
        mma_step_interleaved<T, LAYOUT>(acc, f0a, f0b);
        __builtin_amdgcn_sched_barrier(0);
        // barrier q: K tile q + 1 has landed (read below), and everyone is done with tile q - 1 (its slot is refilled).
        // Half a tile into the iteration: the first tile of a launch does not wait for the second one to land
        asm volatile("s_barrier" ::: "memory");
        const int nxt = (cur + 1 == WS_NST) ? 0 : cur + 1;
        const char* tn = smem + nxt * STAGE_BYTES;  // K tile t + 1 (past the last one: read and never used)
        load_frags_in_use_order<T, LAYOUT>(f0a, f0b, tn, tn + TILE_BYTES, wm, wn, 0);
        mma_step_interleaved<T, LAYOUT>(acc, f1a, f1b);
        __builtin_amdgcn_sched_barrier(0);
        if (LAYOUT == IMT_TN && do_colsum) cs.add_tile(ta);
        cur = nxt;
      }
#else
      for (int t = 0; t < nt; ++t) {
        asm volatile("s_barrier" ::: "memory");
        if (i == 0 && t == 0) IMT_STAMP(ep.trace, 1);
        const char* ta = smem + cur * STAGE_BYTES;
        // (-DIMT_WS_ABLATE=1: fill only.  Compile-time: a run-time branch around the product makes the compiler copy
        // the 64 accumulator registers at every K tile, which is what such a probe then measures)
        if (!(IMT_WS_ABLATE & 1)) compute_tile<T, LAYOUT, IMT_WS_AHEAD>(acc, ta, ta + TILE_BYTES, wm, wn);
        if (LAYOUT == IMT_TN && do_colsum) cs.add_tile(ta);
        cur = (cur + 1 == WS_NST) ? 0 : cur + 1;
      }
#endif
      if (i == 0) IMT_STAMP(ep.trace, 2);
      EpiParams eps = ep;
      if (S > 1) eps.C = reinterpret_cast<float*>(ep.C) + (int64_t)ks * ep.slab_elems;
      if (i == my_tiles - 1) {
        epilogue<T, 512, true>(acc, epi, m0, n0, wm, wn, M, N, eps, alpha);  // producers join in
        if (IMT_LN_TICKET && ep.ln_out) ln_rowblock_tail<T>(ep, epi, m0, lt / nbx, nbx, M, N);
      } else epilogue<T>(acc, epi, m0, n0, wm, wn, M, N, eps, alpha);
      if (i == 0) IMT_STAMP(ep.trace, 3);
      if (i == my_tiles - 1) IMT_STAMP(ep.trace, 4);
      if (colsum_kernel) cs.flush(epi, ep.a_colsum, m0, M, alpha, do_colsum);
    }
  }
}

// ------------------------------------------------------------------------------------------------ 256 x 256 tile kernel
// Every LDS-DMA loop above tops out near 40 GB/s of operand stream per CU, and a 128 x 128 tile needs 32 KiB of it per
// 4.2 MFLOP -- that, not the MFMA rate, is what holds the short-K products (N >= 2048, K = 512: FFN up, its dX, the
// vocabulary projection) at 300-600 TFLOP/s.  A 256 x 256 tile halves the bytes per flop: 8 waves as 2 (m) x 4 (n), each
// 128 x 64 (acc[8][4], 12 fragment reads per 32 MFMAs instead of 8 per 16), operands as four 16-KiB sub-tiles
// (A rows 0-127 / 128-255, B columns 0-127 / 128-255) in the usual swizzled geometry, two 64-KiB stages: waves 0-3
// DMA the A sub-tiles, waves 4-7 the B sub-tiles of K tile t+1 while everyone multiplies tile t; one barrier per K tile.
// The epilogue walks the four 128 x 128 quadrants with all 512 threads (the two owning waves restage, everyone stores).
constexpr int XL_THREADS = 512;
constexpr int XL_STAGE = 4 * TILE_BYTES;
constexpr int XL_LDS = 2 * XL_STAGE;

// One wave's share of TWO neighbouring 16-KiB sub-tiles (operand rows / columns +128): the second sub-tile's source
// offsets are the first's plus a constant, so a wave keeps 4 offset registers whichever operand it streams.
template <typename T> struct DmaPair {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff[4], kadv, delta, wv;
  template <bool KCONTIG> IMT_DEVICE void init(const T* base, int64_t ld, int64_t valid_bytes, int row0, int wave) {
    typedef TileGeom<T, KCONTIG> G;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (int)valid_bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    wv = wave;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = 64 * (4 * i + wave) + lane;
      const int tr = q / G::CPR, pc = q % G::CPR;
      const int c = pc ^ swz<G::RB>(tr);
      if (KCONTIG) voff[i] = (int)((((int64_t)(row0 + tr)) * ld + c * G::EPC) * (int64_t)sizeof(T));
      else         voff[i] = (int)((((int64_t)tr) * ld + row0 + c * G::EPC) * (int64_t)sizeof(T));
    }
    kadv = KCONTIG ? 128 : (int)(G::BK * ld * (int64_t)sizeof(T));
    delta = KCONTIG ? (int)(128 * ld * (int64_t)sizeof(T)) : (int)(128 * sizeof(T));
  }
  // The K advance and the +128-row delta are wave-uniform, but they go into the VECTOR offset: the descriptor's range
  // check (num_records = valid_bytes, which is what turns the rows past M of a ragged last tile into zeros instead of reads
  // past the operand) covers vgpr offset + instruction offset only -- the SGPR `soffset` operand is added AFTER the check
  // (tools/probe_soffset.hip; the round-1 fault of tools/probe_fill.hip was exactly an soffset beyond num_records).
  IMT_DEVICE void issue(char* tiles, int t) const {
    const int wave = __builtin_amdgcn_readfirstlane(wv);
    const int adv = __builtin_amdgcn_readfirstlane(t * kadv), d = __builtin_amdgcn_readfirstlane(delta);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(tiles + h * TILE_BYTES + (4 * i + wave) * 1024),
                                                 16, voff[i] + (adv + h * d), 0, 0, 0);
  }
};

// Epilogue of a FULL 256 x 256 tile straight from the accumulators: lane (lr, lg) owns 4 consecutive n of row 16i + lr in
// each of its wave's 8 x 4 MFMA tiles, so bias / aux / C accesses are 8-byte (bf16) vectors, 32 B contiguous per row and
// instruction, a full 128-B line per row over j = 0..3.  No LDS pass, no barrier, and NO run-time branch: the epilogue
// kind is a template parameter, because 32 unrolled copies of a body that branches over every option (GELU, GELU',
// dropout, residual, ...) are ~250 KB of code that a wave walks once, taking an instruction-cache miss at every skipped
// block -- measured 12-18 us per tile (IMT_GEMM_TRACE), more than the whole K loop.  Kinds outside the table below and
// ragged edge tiles take the restaged path.
template <typename T, int AUX, bool C_F32, bool RAGGED_M>
IMT_DEVICE void epilogue_xl_direct(const f32x4 (&acc)[8][4], int mw, int nw, int M, const EpiParams& ep, float alpha) {
  typedef typename Vec4<T>::type raw_t;
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
  const T* bias = reinterpret_cast<const T*>(ep.bias);
  T* aux = reinterpret_cast<T*>(ep.aux);
  const int nl = nw + 4 * lg;
  f32x4 bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = bias ? Vec4<T>::load(bias + nl + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
  raw_t zr[2][4];
  auto fetch = [&](int i, int b) {
    if (AUX == IMT_AUX_DGELU) {
      // ragged M: clamped loads, predicated stores
      const int64_t m = RAGGED_M ? min((int64_t)(mw + 16 * i + lr), (int64_t)M - 1) : (int64_t)(mw + 16 * i + lr);
#pragma unroll
      for (int j = 0; j < 4; ++j) zr[b][j] = Vec4<T>::load_raw(aux + m * ep.ldaux + nl + 16 * j);
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i + 1 < 8) fetch(i + 1, (i + 1) & 1);  // requested before this group's stores: the wait below never covers a store
    const int64_t m = mw + 16 * i + lr;
    const bool live = !RAGGED_M || m < M;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nl + 16 * j;
      f32x4 v = acc[i][j] * alpha + bv[j];
      if (AUX == IMT_AUX_GELU_FWD) {
        if (live) Vec4<T>::store(aux + m * ep.ldaux + n, v);
        v = gelu_erf4(v);
      } else if (AUX == IMT_AUX_DGELU) {
        const f32x4 z = Vec4<T>::cvt(zr[i & 1][j]);
        v *= gelu_erf_grad4(z);
      }
      if (live && !(ep.dbg & 128)) {  // dbg bit 7 (tuning only): no output stores -- what the stores and their write-back cost
        if (C_F32) Vec4<float>::store(reinterpret_cast<float*>(ep.C) + m * ep.ldc + n, v);
        else if (ep.dbg & 256) Vec4<T>::store_wt(reinterpret_cast<T*>(ep.C) + m * ep.ldc + n, v);  // bit 8: write-through stores
        else       Vec4<T>::store(reinterpret_cast<T*>(ep.C) + m * ep.ldc + n, v);
      }
    }
  }
}

template <typename T, int LAYOUT, int HALF = -1>
IMT_DEVICE void compute_tile_xl(f32x4 (&acc)[8][4], const char* ta, const char* tb, int wn) {
  constexpr bool A_KC = (LAYOUT != IMT_TN), B_KC = (LAYOUT == IMT_NT);
  typedef TileGeom<T, A_KC> GA;
  typedef TileGeom<T, B_KC> GB;
  typedef typename Frag<T>::type frag_t;
  constexpr int KSTEP = Frag<T>::KSTEP;
  constexpr int NSTEP = GA::BK / KSTEP;  // HALF = 0 / 1: the first / second half of the tile's K steps only
  constexpr int S0 = HALF == 1 ? NSTEP / 2 : 0, S1 = HALF == 0 ? NSTEP / 2 : NSTEP;
#pragma unroll
  for (int s = S0; s < S1; ++s) {
    frag_t fa[8], fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (B_KC) fb[j] = lds_frag_kcontig<T, GB::RB>(tb, wn + 16 * j, 4 * s);
      else      fb[j] = KStrided<T, GB::RB>::load(tb, s * KSTEP, wn + 16 * j);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (A_KC) fa[i] = lds_frag_kcontig<T, GA::RB>(ta, 16 * i, 4 * s);
      else      fa[i] = KStrided<T, GA::RB>::load(ta, s * KSTEP, 16 * i);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) mma16(acc[i][j], fb[j], fa[i]);
    // keep the next K step's 12 fragment reads below this step's MFMAs: hoisted, they push the 128 accumulator
    // registers + 2 x 48 fragment registers past the 256 a wave gets at two waves per SIMD (spills in the loop)
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <typename T, int LAYOUT>
__global__ __launch_bounds__(XL_THREADS) void gemm_xl_kernel(const T* __restrict__ A, int64_t lda, int64_t a_bytes,
                                                             const T* __restrict__ B, int64_t ldb, int64_t b_bytes, int M, int N,
                                                             int K, EpiParams ep) {
  constexpr bool A_KC = (LAYOUT != IMT_TN), B_KC = (LAYOUT == IMT_NT);
  constexpr int BK = TileGeom<T, A_KC>::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nbx = (N + 255) / 256, nby = (M + 255) / 256;
  const int bid = imt_xcd_block(blockIdx.x, nbx * nby);
  const int m0 = (bid / nbx) * 256, n0 = (bid % nbx) * 256;
  // split-K slab mode (gridDim.y > 1): this workgroup multiplies K tiles [kt0, kt0 + nt) into its own fp32 slab
  const int nt_all = (K + BK - 1) / BK;  // whole K tiles for NT / NN (host-checked); TN: a ragged last tile reads zeros (range check)
  const int per = (nt_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int kt0 = (int)blockIdx.y * per;
  const int nt = max(0, min(per, nt_all - kt0));
  if (gridDim.y > 1) ep.C = reinterpret_cast<float*>(ep.C) + (int64_t)blockIdx.y * ep.slab_elems;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wmi = wave >> 2, wni = wave & 3, wn = (wni & 1) * 64;
  const bool loads_a = wave < 4;

  IMT_STAMP(ep.trace, 0);
  DmaPair<T> dma;
  if (loads_a) dma.template init<A_KC>(A, lda, a_bytes, m0, wave);
  else         dma.template init<B_KC>(B, ldb, b_bytes, n0, wave - 4);
  auto issue = [&](int slot, int t) { dma.issue(smem + slot * XL_STAGE + (loads_a ? 0 : 2 * TILE_BYTES), kt0 + t); };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (no K-order rotation: measured no gain here, and the same K order keeps results bit-identical to the other kernels)
  // TN weight gradients: the fused bias gradient (column sums of the K-strided A tiles) -- threads 0-255 own the first
  // 128-column sub-tile, 256-511 the second; only the workgroups of the first tile column take part
  ColSum<T> cs;
  cs.clear();
  const bool do_colsum = (LAYOUT == IMT_TN) && ep.a_colsum && n0 == 0;
  const int half = threadIdx.x >> 8, ht = threadIdx.x & 255;
  if (nt > 0) issue(0, 0);
  for (int t = 0; t < nt; ++t) {
    // my eight pieces of tile t have landed; the barrier publishes everyone's and proves that the other stage (read
    // while multiplying tile t-1) is free for the DMA of tile t+1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    // Issuing a wave's eight 1-KiB DMA pieces takes it ~0.4 us during which it multiplies nothing (one wave sustains
    // ~20 GB/s of LDS-DMA, profiles/r01_lds_fill_probe.txt).  Waves w and w+4 share a SIMD: the first issues before its
    // MFMAs, the second between its two K steps, so each SIMD always has one wave multiplying.
    const bool issue_now = (t + 1 < nt) && !(ep.dbg & 2);
    if (issue_now && (wave < 4 || (ep.dbg & 64))) issue((t + 1) & 1, t + 1);
    if (t == 0) IMT_STAMP(ep.trace, 1);
    const char* st = smem + (t & 1) * XL_STAGE;
    if (!(ep.dbg & 1)) compute_tile_xl<T, LAYOUT, 0>(acc, st + wmi * TILE_BYTES, st + (2 + (wni >> 1)) * TILE_BYTES, wn);
    if (issue_now && wave >= 4 && !(ep.dbg & 64)) issue((t + 1) & 1, t + 1);
    if (!(ep.dbg & 1)) compute_tile_xl<T, LAYOUT, 1>(acc, st + wmi * TILE_BYTES, st + (2 + (wni >> 1)) * TILE_BYTES, wn);
    if (LAYOUT == IMT_TN && do_colsum) cs.add_tile(st + half * TILE_BYTES, ht);
  }
  const float alpha = ep.alpha_dev ? ep.alpha * ep.alpha_dev[0] : ep.alpha;
  IMT_STAMP(ep.trace, 2);
  if (n0 + 256 <= N && !ep.atomic && !ep.accumulate && !ep.resid && !ep.drop_thresh && !(ep.dbg & 8)) {
    const int mw = m0 + 128 * wmi, nw = n0 + 64 * wni;
    bool done = true;
    const bool rag = m0 + 256 > M;
    if (ep.aux_mode == IMT_AUX_NONE && !ep.c_f32) {
      if (rag) epilogue_xl_direct<T, IMT_AUX_NONE, false, true>(acc, mw, nw, M, ep, alpha);
      else     epilogue_xl_direct<T, IMT_AUX_NONE, false, false>(acc, mw, nw, M, ep, alpha);
    } else if (ep.aux_mode == IMT_AUX_NONE) {
      if (rag) epilogue_xl_direct<T, IMT_AUX_NONE, true, true>(acc, mw, nw, M, ep, alpha);
      else     epilogue_xl_direct<T, IMT_AUX_NONE, true, false>(acc, mw, nw, M, ep, alpha);
    } else if (ep.aux_mode == IMT_AUX_GELU_FWD && !ep.c_f32) {
      if (rag) epilogue_xl_direct<T, IMT_AUX_GELU_FWD, false, true>(acc, mw, nw, M, ep, alpha);
      else     epilogue_xl_direct<T, IMT_AUX_GELU_FWD, false, false>(acc, mw, nw, M, ep, alpha);
    } else if (ep.aux_mode == IMT_AUX_DGELU && !ep.c_f32) {
      if (rag) epilogue_xl_direct<T, IMT_AUX_DGELU, false, true>(acc, mw, nw, M, ep, alpha);
      else     epilogue_xl_direct<T, IMT_AUX_DGELU, false, false>(acc, mw, nw, M, ep, alpha);
    }
    else done = false;
    if (done) {
      IMT_STAMP(ep.trace, 3);
      if (LAYOUT == IMT_TN && do_colsum) cs.flush(smem + half * 16384, ep.a_colsum, m0 + 128 * half, M, alpha, true, ht);
      return;
    }
  }
  // ragged edge tiles: the bounds-checked restaged epilogue, one 128 x 128 quadrant at a time
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    const int qm = q >> 1, qn = q & 1;
    const int mq = m0 + 128 * qm, nq = n0 + 128 * qn;
    if (mq >= M || nq >= N) continue;  // uniform over the workgroup
    if (wmi == qm && (wni >> 1) == qn) epilogue<T, XL_THREADS, true, 128>(acc, smem, mq, nq, 0, wn, M, N, ep, alpha);
    else                               epilogue<T, XL_THREADS, false, 128>(acc, smem, mq, nq, 0, wn, M, N, ep, alpha);
  }
  if (LAYOUT == IMT_TN && do_colsum) cs.flush(smem + half * 16384, ep.a_colsum, m0 + 128 * half, M, alpha, true, ht);
}

// ------------------------------------------------------------------------------------------------ grouped weight gradients
// All weight-gradient GEMMs of one transformer layer (dW = dy^T x, K = tokens) in ONE launch: each has only 16-64
// output tiles, so separately they either idle most CUs or need split-K atomics; together they are ~one tile per
// CU with the full K per block (long steady-state LDS-DMA pipeline, direct fp32 accumulate, no atomics).
constexpr int MAX_GROUP = 8;
struct GroupProblem {
  const void* A; const void* B; float* C; float* a_colsum;
  int64_t lda, ldb, ldc, a_bytes, b_bytes;
  int M, N, K, tile_start;
};
struct GroupArgs { int count; int total_tiles; float alpha; int alpha_pad_order /* tuning: 1 = plain workgroup order */; GroupProblem p[MAX_GROUP]; };

// Wave-specialised: waves 0-3 multiply (one per SIMD), waves 4-7 only issue the LDS-DMA of the tiles ahead.  Issuing
// a 1-KiB DMA piece costs a wave ~100-200 cycles (MI355X_MICROARCH "LDS-DMA piece issue cost"); done by the MFMA waves
// themselves it serialises with their tile products (measured: loads-only 35 us + mfma-only 35 us -> 47 us, not
// max()), done by partner waves on the same SIMDs it overlaps.  One raw s_barrier per K tile: producers first wait
// (counted vmcnt) for THEIR pieces of tile t, the barrier publishes tile t to the consumers and tells the producers
// that the consumers are done with the buffer the next DMA overwrites.
constexpr int GROUP_NST = IMT_GROUP_NST;       // 4: 128 KiB ring, three tiles in flight
constexpr int GROUP_THREADS = 512;

// Consumer side of the grouped weight-gradient kernel: software pipeline over k-steps, across tile boundaries -- the LDS
// reads of a k-step are issued one by one in the shadows of the previous k-step's MFMAs (one multiplying wave per SIMD:
// nobody else hides an LDS round trip, and a burst of reads in front of the MFMAs leaves the matrix pipe idle while they
// issue).  Barrier t tells the consumers that tile t + 1 has landed, so the first fragments of tile t + 1 are read while
// tile t is still being multiplied.
// CS: this workgroup also owns the bias gradient of its 128 rows of dy (column sums of the A tiles).  They ride on the
// matrix pipe: D = ones x A-fragment is a 16 x 16 tile whose every row holds the column sums of those 16 columns, and the
// A fragments are in registers already -- two more MFMAs per k-step and wave (the two waves that share an A strip take two
// fragments each) instead of a second pass over the LDS tile with ~100 VALU instructions per K tile, which made these
// workgroups the stragglers of the launch (decoder layer: 88 -> 70 us without bias gradients).
template <typename T> IMT_DEVICE typename Frag<T>::type ones_frag();
template <> IMT_DEVICE f32x4 ones_frag<float>() { return f32x4{1.f, 1.f, 1.f, 1.f}; }
template <> IMT_DEVICE bf16x8 ones_frag<bf16_t>() {
  const bf16_t o = (bf16_t)1.0f;
  return bf16x8{o, o, o, o, o, o, o, o};
}

template <typename T, bool CS, bool LO>
IMT_DEVICE void grouped_consumer_loop(f32x4 (&acc)[4][4], f32x4 (&bacc)[2], const char* smem, int nt, int wm, int wn) {
  typedef typename Frag<T>::type frag_t;
  constexpr int MPF = sizeof(T) == 2 ? 1 : 4;  // MFMA instructions per fragment pair
  frag_t f0a[4], f0b[4], f1a[4], f1b[4];
  const frag_t ones = ones_frag<T>();
  // (LO: which two of the wave's four A fragments it sums -- a template parameter, not a branch: a branch inside the loop
  // splits the scheduling region and the compiler bunches the reads again)
  asm volatile("s_barrier" ::: "memory");  // tile 0 landed
  load_frags_in_use_order<T, IMT_TN>(f0a, f0b, smem, smem + TILE_BYTES, wm, wn, 0);
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_barrier" ::: "memory");
    const char* ta = smem + cur * STAGE_BYTES;
    load_frags_in_use_order<T, IMT_TN>(f1a, f1b, ta, ta + TILE_BYTES, wm, wn, 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) if (!(IMT_WS_ABLATE & 1)) mma16(acc[i][j], f0b[j], f0a[i]);
    if (CS) { mma16(bacc[0], ones, f0a[LO ? 0 : 2]); mma16(bacc[1], ones, f0a[LO ? 1 : 3]); }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, MPF, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // one DS read
    }
    __builtin_amdgcn_sched_barrier(0);
    const int nxt = (cur + 1 == GROUP_NST) ? 0 : cur + 1;
    const char* tn = smem + nxt * STAGE_BYTES;  // tile t + 1 (landed: barrier t); past the last tile: read and never used
    load_frags_in_use_order<T, IMT_TN>(f0a, f0b, tn, tn + TILE_BYTES, wm, wn, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) if (!(IMT_WS_ABLATE & 1)) mma16(acc[i][j], f1b[j], f1a[i]);
    if (CS) { mma16(bacc[0], ones, f1a[LO ? 0 : 2]); mma16(bacc[1], ones, f1a[LO ? 1 : 3]); }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, MPF, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    cur = nxt;
  }
}

template <typename T>
__global__ __launch_bounds__(GROUP_THREADS) void gemm_grouped_tn_kernel(GroupArgs g) {
  constexpr int BK = TileGeom<T, false>::BK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-affine tile order: hardware workgroup b runs on XCD b % 8; give each XCD a CONTIGUOUS run of the group's tile list
  // (row-major per problem), i.e. a few whole tile rows -- its workgroups then share dy / x operand strips through that
  // XCD's L2.  With the plain order every XCD saw one column and ~24 different rows: 549 MB of fabric traffic per launch
  // against 160 MB algorithmic, and the kernel ran at the fabric's ~6 TB/s.
  const int bid = (g.alpha_pad_order == 0) ? imt_xcd_block(blockIdx.x, g.total_tiles) : (int)blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_GROUP; ++i)
    if (i < g.count && bid >= g.p[i].tile_start) pi = i;
  const GroupProblem& P = g.p[pi];
  const int M = P.M, N = P.N, K = P.K;
  const int nbx = (N + BN - 1) / BN;
  const int local = bid - P.tile_start;
  const int m0 = (local / nbx) * BM, n0 = (local % nbx) * BN;
  const int nt = (K + BK - 1) / BK;  // a ragged last K tile (token counts are arbitrary) reads zeros past the operands' valid bytes
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool consumer = wave < 4;
  const int wm = ((wave & 3) >> 1) * 64, wn = (wave & 1) * 64;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  ColSum<T> cs;
  cs.clear();
  f32x4 bacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // bias gradient (column sums of dy), IMT_GROUP_RUNAHEAD
  const bool do_colsum = P.a_colsum && n0 == 0;

  if (!consumer) {
    // ------------------------------------------------------------ producer waves
    Dma<T, false> da; Dma<T, false> db;
    da.init(reinterpret_cast<const T*>(P.A), P.lda, P.a_bytes, m0, 0, wave - 4);
    db.init(reinterpret_cast<const T*>(P.B), P.ldb, P.b_bytes, n0, 0, wave - 4);
#pragma unroll
    for (int s0 = 0; s0 < GROUP_NST - 1; ++s0)
      if (s0 < nt) { da.issue(smem + s0 * STAGE_BYTES, s0); db.issue(smem + s0 * STAGE_BYTES + TILE_BYTES, s0); }
    int cur = 0;
#if IMT_GROUP_RUNAHEAD
    // barrier t tells the consumers that tile t + 1 has landed (they read its first fragments while still multiplying
    // tile t), and the producers that the consumers are done with tile t - 1, whose slot takes tile t + 3
    {
      const int newer0 = min(GROUP_NST - 2, nt - 1);
      if (newer0 >= 3)      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if (newer0 == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (newer0 == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else                  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");  // tile 0 landed
    }
    for (int t = 0; t < nt; ++t) {
      const int newer = max(0, min(nt - 1, t + GROUP_NST - 2) - (t + 1));  // tiles issued after tile t + 1
      if (newer >= 2)      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if (t + GROUP_NST - 1 < nt && !(IMT_WS_ABLATE & 2)) {
        const int nxt = (cur == 0) ? GROUP_NST - 1 : cur - 1;
        da.issue(smem + nxt * STAGE_BYTES, t + GROUP_NST - 1);
        db.issue(smem + nxt * STAGE_BYTES + TILE_BYTES, t + GROUP_NST - 1);
      }
      cur = (cur + 1 == GROUP_NST) ? 0 : cur + 1;
    }
#else
    for (int t = 0; t < nt; ++t) {
      const int newer = min(GROUP_NST - 2, nt - 1 - t);
      if (newer >= 2)      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (newer == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if (t + GROUP_NST - 1 < nt) {
        const int nxt = (cur == 0) ? GROUP_NST - 1 : cur - 1;
        da.issue(smem + nxt * STAGE_BYTES, t + GROUP_NST - 1);
        db.issue(smem + nxt * STAGE_BYTES + TILE_BYTES, t + GROUP_NST - 1);
      }
      cur = (cur + 1 == GROUP_NST) ? 0 : cur + 1;
    }
#endif
  } else {
    // ------------------------------------------------------------ consumer waves
    int cur = 0;
#if IMT_GROUP_RUNAHEAD
    if (!do_colsum)   grouped_consumer_loop<T, false, false>(acc, bacc, smem, nt, wm, wn);
    else if (wn == 0) grouped_consumer_loop<T, true, true>(acc, bacc, smem, nt, wm, wn);
    else              grouped_consumer_loop<T, true, false>(acc, bacc, smem, nt, wm, wn);
#else
    for (int t = 0; t < nt; ++t) {
      asm volatile("s_barrier" ::: "memory");
      const char* ta = smem + cur * STAGE_BYTES;
      compute_tile<T, IMT_TN>(acc, ta, ta + TILE_BYTES, wm, wn);
      if (do_colsum) cs.add_tile(ta);
      cur = (cur + 1 == GROUP_NST) ? 0 : cur + 1;
    }
#endif
  }
  EpiParams ep;
  ep.C = P.C; ep.ldc = P.ldc; ep.c_f32 = 1; ep.accumulate = 1;
  ep.bias = nullptr; ep.resid = nullptr; ep.ldr = 0; ep.aux = nullptr; ep.ldaux = 0; ep.aux_mode = IMT_AUX_NONE;
  ep.atomic = 0; ep.alpha = g.alpha; ep.alpha_dev = nullptr; ep.inv_keep = 1.f; ep.drop_thresh = 0; ep.seed = 0;
  ep.a_colsum = P.a_colsum; ep.dbg = 0; ep.trace = nullptr;
  // one tile per workgroup: the epilogue is fully exposed, so all 8 waves share it (the producers have nothing left to do)
  if (consumer) epilogue<T, 512, true>(acc, smem, m0, n0, wm, wn, M, N, ep, g.alpha);
  else epilogue<T, 512, false>(acc, smem, m0, n0, wm, wn, M, N, ep, g.alpha);
#if IMT_GROUP_RUNAHEAD
  if (do_colsum && consumer && (threadIdx.x & 63) < 16) {
    // every row of the 16 x 16 tile holds the same sums: lanes 0-15 (row group 0) publish element 0; one contributor per column
    const int mb = m0 + wm + ((wn == 0) ? 0 : 32) + (threadIdx.x & 15);
    if (mb < M)      atomicAdd(P.a_colsum + mb, bacc[0][0] * g.alpha);
    if (mb + 16 < M) atomicAdd(P.a_colsum + mb + 16, bacc[1][0] * g.alpha);
  }
#else
  if (do_colsum) cs.flush(smem, P.a_colsum, m0, M, g.alpha, consumer);
#endif
}

// ------------------------------------------------------------------------------------------------ host side
int64_t view_bytes(int rows, int64_t ld, int inner, int es) { return rows > 0 ? ((int64_t)(rows - 1) * ld + inner) * es : 0; }

template <typename T, int LAYOUT>
int launch(const imt_gemm_args* a, const EpiParams& ep, int splits, int k_per_split, int variant, hipStream_t st) {
  const int nbx = imt_cdiv(a->N, BN), nby = imt_cdiv(a->M, BM);
  dim3 grid(nbx * nby, splits);
  static bool attr_set = false;
  auto kern = gemm_kernel<T, LAYOUT>;
  auto kpipe = gemm_pipe_kernel<T, LAYOUT, 3>;
  auto kpipe4 = gemm_pipe_kernel<T, LAYOUT, 4>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kpipe), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kpipe4), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE_BYTES);
    attr_set = true;
  }
  // profiler kind == one kernel symbol: gemm_<variant>_<dtype>_<layout> (variant: dbuf = gemm_kernel, dma = gemm_pipe_kernel,
  // sbuf = gemm_sb_kernel, ws = gemm_ws_kernel) so that rocprofv3's per-symbol averages can be compared one to one
  static const char* const kinds[7][2][3] = {
      {{"", "", ""}, {"", "", ""}},
      {{"gemm_dbuf_f32_nt", "gemm_dbuf_f32_nn", "gemm_dbuf_f32_tn"}, {"gemm_dbuf_bf16_nt", "gemm_dbuf_bf16_nn", "gemm_dbuf_bf16_tn"}},
      {{"gemm_dma_f32_nt", "gemm_dma_f32_nn", "gemm_dma_f32_tn"}, {"gemm_dma_bf16_nt", "gemm_dma_bf16_nn", "gemm_dma_bf16_tn"}},
      {{"gemm_sbuf_f32_nt", "gemm_sbuf_f32_nn", "gemm_sbuf_f32_tn"}, {"gemm_sbuf_bf16_nt", "gemm_sbuf_bf16_nn", "gemm_sbuf_bf16_tn"}},
      {{"gemm_dma_f32_nt", "gemm_dma_f32_nn", "gemm_dma_f32_tn"}, {"gemm_dma_bf16_nt", "gemm_dma_bf16_nn", "gemm_dma_bf16_tn"}},
      {{"gemm_ws_f32_nt", "gemm_ws_f32_nn", "gemm_ws_f32_tn"}, {"gemm_ws_bf16_nt", "gemm_ws_bf16_nn", "gemm_ws_bf16_tn"}},
      {{"gemm_xl_f32_nt", "gemm_xl_f32_nn", "gemm_xl_f32_tn"}, {"gemm_xl_bf16_nt", "gemm_xl_bf16_nn", "gemm_xl_bf16_tn"}}};
  const double es = sizeof(T), esc = ep.c_f32 ? 4.0 : es;
  const char* kind = kinds[variant >= 1 && variant <= 6 ? variant : 1][sizeof(T) == 2][LAYOUT];
  if (imt_prof_enabled() && getenv("IMT_PROF_SHAPES")) kind = imt_prof_intern(kind, a->M, a->N, a->K);
  ImtProfScope prof(kind, 2.0 * a->M * a->N * a->K,
                    ((double)a->M * a->K + (double)a->N * a->K) * es + (double)a->M * a->N * esc, st);
  const T* A = reinterpret_cast<const T*>(a->A);
  const T* B = reinterpret_cast<const T*>(a->B);
  if (variant == 6) {
    static bool xl_attr = false;
    auto kxl = gemm_xl_kernel<T, LAYOUT>;
    if (!xl_attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kxl), hipFuncAttributeMaxDynamicSharedMemorySize, XL_LDS); xl_attr = true; }
    const int64_t a_bytes = (LAYOUT == IMT_TN) ? view_bytes(a->K, a->lda, a->M, sizeof(T)) : view_bytes(a->M, a->lda, a->K, sizeof(T));
    const int64_t b_bytes = (LAYOUT == IMT_NT) ? view_bytes(a->N, a->ldb, a->K, sizeof(T)) : view_bytes(a->K, a->ldb, a->N, sizeof(T));
    const int nwg = imt_cdiv(a->M, 256) * imt_cdiv(a->N, 256);
    ImtTrace tr("gemm_xl", splits == 1 ? nwg : 0, st);  // IMT_TRACE=gemm_xl: phases = first K tile landed | K loop | epilogue
    EpiParams ept = ep;
    ept.trace = tr.dev;
    hipLaunchKernelGGL(kxl, dim3(nwg, splits), dim3(XL_THREADS), XL_LDS, st, A, a->lda, a_bytes, B, a->ldb, b_bytes, a->M, a->N, a->K, ept);
    if (tr.dev) fprintf(stderr, "[gemm_xl %s %dx%dx%d]\n", kind, a->M, a->N, a->K);
  } else if (variant == 5) {
    static bool ws_attr = false;
    auto kws = gemm_ws_kernel<T, LAYOUT>;
    if (!ws_attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kws), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS); ws_attr = true; }
    const int64_t a_bytes = (LAYOUT == IMT_TN) ? view_bytes(a->K, a->lda, a->M, sizeof(T)) : view_bytes(a->M, a->lda, a->K, sizeof(T));
    const int64_t b_bytes = (LAYOUT == IMT_NT) ? view_bytes(a->N, a->ldb, a->K, sizeof(T)) : view_bytes(a->K, a->ldb, a->N, sizeof(T));
    const int tiles = nbx * nby * (ep.splits > 1 ? ep.splits : 1);
    ImtTrace tr("gemm_ws", tiles < 256 ? tiles : 256, st);
    EpiParams ept = ep;
    ept.trace = tr.dev;
    hipLaunchKernelGGL(kws, dim3(tiles < 256 ? tiles : 256), dim3(WS_THREADS), WS_LDS, st, A, a->lda, a_bytes, B, a->ldb, b_bytes, a->M, a->N, a->K, ept);
    if (tr.dev) fprintf(stderr, "[gemm_ws %s %dx%dx%d bias %d resid %d drop %d aux %d acc %d]\n", kind, a->M, a->N, a->K, ep.bias != nullptr, ep.resid != nullptr, ep.drop_thresh != 0, ep.aux_mode, ep.accumulate);
  } else if (variant == 3) {
    hipLaunchKernelGGL((gemm_sb_kernel<T, LAYOUT>), grid, dim3(NTHREADS), STAGE_BYTES, st, A, a->lda, B, a->ldb, a->M, a->N, a->K, k_per_split, ep);
  } else if (variant == 2 || variant == 4) {
    const int64_t a_bytes = (LAYOUT == IMT_TN) ? view_bytes(a->K, a->lda, a->M, sizeof(T)) : view_bytes(a->M, a->lda, a->K, sizeof(T));
    const int64_t b_bytes = (LAYOUT == IMT_NT) ? view_bytes(a->N, a->ldb, a->K, sizeof(T)) : view_bytes(a->K, a->ldb, a->N, sizeof(T));
    if (variant == 4)
      hipLaunchKernelGGL(kpipe4, grid, dim3(NTHREADS), 4 * STAGE_BYTES, st, A, a->lda, a_bytes, B, a->ldb, b_bytes, a->M, a->N, a->K, k_per_split, ep);
    else
      hipLaunchKernelGGL(kpipe, grid, dim3(NTHREADS), 3 * STAGE_BYTES, st, A, a->lda, a_bytes, B, a->ldb, b_bytes, a->M, a->N, a->K, k_per_split, ep);
  } else {
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), 2 * STAGE_BYTES, st, A, a->lda, B, a->ldb, a->M, a->N, a->K, k_per_split, ep);
  }
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

template <typename T> int dispatch(const imt_gemm_args* a, const EpiParams& ep, int splits, int kps, int variant, hipStream_t st) {
  switch (a->layout) {
    case IMT_NT: return launch<T, IMT_NT>(a, ep, splits, kps, variant, st);
    case IMT_NN: return launch<T, IMT_NN>(a, ep, splits, kps, variant, st);
    case IMT_TN: return launch<T, IMT_TN>(a, ep, splits, kps, variant, st);
  }
  imt_set_error("imt_gemm: bad layout %d", a->layout);
  return IMT_ERR_BAD_ARG;
}

// C[m][n] = (accumulate ? C[m][n] : 0) + sum_s slab[s][m][n]      (split-K slab mode of the 256-tile kernel)
template <typename TC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int64_t slab_elems, int splits,
                                                            TC* __restrict__ C, int64_t ldc, int M, int N, int accumulate) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // 4-column group index
  const int n4 = N >> 2;
  if (q >= (int64_t)M * n4) return;
  const int m = (int)(q / n4), n = (int)(q % n4) * 4;
  f32x4 v = Vec4<float>::load(slabs + (int64_t)m * N + n);
  for (int sidx = 1; sidx < splits; ++sidx) v += Vec4<float>::load(slabs + sidx * slab_elems + (int64_t)m * N + n);
  TC* c = C + (int64_t)m * ldc + n;
  if (accumulate) v += Vec4<TC>::load(c);
  Vec4<TC>::store(c, v);
}

// Split-K slab mode of the persistent kernel (imt_gemm_args.splitk_ws): C = epilogue(alpha * sum_s slab[s]) -- the slabs are
// summed in split order whatever order they were written in, then bias / GELU (aux <- pre-activation) / GELU' (aux) /
// dropout (element index m * N + n, as the GEMM epilogues) / residual / accumulate, C of the operand type or fp32.
template <typename T>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ slabs, int64_t slab_elems, int splits, int M, int N,
                                                              EpiParams ep) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;  // 4-column group index (N % 4 == 0)
  const int n4 = N >> 2;
  if (q >= (int64_t)M * n4) return;
  const int m = (int)(q / n4), n = (int)(q % n4) * 4;
  f32x4 v = Vec4<float>::load(slabs + (int64_t)m * N + n);
  for (int sidx = 1; sidx < splits; ++sidx) v += Vec4<float>::load(slabs + sidx * slab_elems + (int64_t)m * N + n);
  const float alpha = ep.alpha_dev ? ep.alpha * ep.alpha_dev[0] : ep.alpha;
  v *= alpha;
  const T* bias = reinterpret_cast<const T*>(ep.bias);
  const T* resid = reinterpret_cast<const T*>(ep.resid);
  T* aux = reinterpret_cast<T*>(ep.aux);
  if (bias) v += Vec4<T>::load(bias + n);
  if (ep.aux_mode == IMT_AUX_GELU_FWD) {
    Vec4<T>::store(aux + (int64_t)m * ep.ldaux + n, v);
    v = gelu_erf4(v);
  } else if (ep.aux_mode == IMT_AUX_DGELU) {
    v *= gelu_erf_grad4(Vec4<T>::load(aux + (int64_t)m * ep.ldaux + n));
  }
  if (ep.drop_thresh) dropout_apply4(v, ep.seed, (uint64_t)m * (uint64_t)N + (uint64_t)n, ep.drop_thresh, ep.inv_keep);
  if (resid) v += Vec4<T>::load(resid + (int64_t)m * ep.ldr + n);
  if (ep.c_f32) {
    float* c = reinterpret_cast<float*>(ep.C) + (int64_t)m * ep.ldc + n;
    if (ep.accumulate) v += Vec4<float>::load(c);
    Vec4<float>::store(c, v);
  } else {
    T* c = reinterpret_cast<T*>(ep.C) + (int64_t)m * ep.ldc + n;
    if (ep.accumulate) v += Vec4<T>::load(c);
    Vec4<T>::store(c, v);
  }
}

// The same with the LayerNorm of the finished row behind it (dense + bias + dropout + residual + LayerNorm of BertSelfOutput /
// BertOutput, src/bert_seq2seq.py:84-90,139-143, for few-row batches): one wave per row sums the slabs, applies the epilogue,
// stores the LayerNorm INPUT rounded to T (what imt_layernorm_bwd reads) and normalises those rounded values with the
// arithmetic of ln_fwd_kernel (rowops.hip) -- bit-identical to the epilogue launch + LayerNorm launch pair it replaces.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void splitk_epilogue_ln_kernel(const float* __restrict__ slabs, int64_t slab_elems, int splits, int M, int N,
                                                                 EpiParams ep) {
  const int lane = threadIdx.x & 63;
  const int row = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (row >= M) return;
  const T* bias = reinterpret_cast<const T*>(ep.bias);
  const T* resid = reinterpret_cast<const T*>(ep.resid);
  const T* gamma = reinterpret_cast<const T*>(ep.ln_gamma);
  const T* beta = reinterpret_cast<const T*>(ep.ln_beta);
  const float alpha = ep.alpha_dev ? ep.alpha * ep.alpha_dev[0] : ep.alpha;
  f32x4 v[NCH], gv[NCH], bv[NCH];
  // every load up front, unconditionally (columns past N read a clamped address and are masked where used)
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = min(lane * 4 + i * 256, N - 4);
    v[i] = Vec4<float>::load(slabs + (int64_t)row * N + c);
    gv[i] = Vec4<T>::load(gamma + c);
    bv[i] = Vec4<T>::load(beta + c);
  }
  for (int sidx = 1; sidx < splits; ++sidx)
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] += Vec4<float>::load(slabs + sidx * slab_elems + (int64_t)row * N + min(lane * 4 + i * 256, N - 4));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 4 + i * 256, cc = min(c, N - 4);
    v[i] *= alpha;
    if (bias) v[i] += Vec4<T>::load(bias + cc);
    if (ep.drop_thresh) dropout_apply4(v[i], ep.seed, (uint64_t)row * (uint64_t)N + (uint64_t)cc, ep.drop_thresh, ep.inv_keep);
    if (resid) v[i] += Vec4<T>::load(resid + (int64_t)row * ep.ldr + cc);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = to_f32<T>(from_f32<T>(v[i][e]));
    if (c < N) {
      Vec4<T>::store(reinterpret_cast<T*>(ep.C) + (int64_t)row * ep.ldc + c, v[i]);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(s) / (float)N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (lane * 4 + i * 256 < N) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float t = v[i][e] - mean; q += t * t; }
    }
  const float var = wave_sum(q) / (float)N;
  const float rstd = 1.0f / sqrtf(var + ep.ln_eps);
  if (lane == 0) {
    if (ep.ln_mean) ep.ln_mean[row] = mean;
    if (ep.ln_rstd) ep.ln_rstd[row] = rstd;
  }
  T* yr = reinterpret_cast<T*>(ep.ln_out) + (int64_t)row * ep.ld_ln;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane * 4 + i * 256;
    if (c < N) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gv[i][e] + bv[i][e];
      Vec4<T>::store(yr + c, o);
    }
  }
}

template <typename T>
int launch_splitk_epilogue_ln(const float* slabs, int64_t slab_elems, int S, int M, int N, const EpiParams& ep, hipStream_t st) {
  const dim3 grid(imt_cdiv(M, 4));
  switch (imt_cdiv(N, 256)) {
    case 1: hipLaunchKernelGGL((splitk_epilogue_ln_kernel<T, 1>), grid, dim3(256), 0, st, slabs, slab_elems, S, M, N, ep); break;
    case 2: hipLaunchKernelGGL((splitk_epilogue_ln_kernel<T, 2>), grid, dim3(256), 0, st, slabs, slab_elems, S, M, N, ep); break;
    case 3: hipLaunchKernelGGL((splitk_epilogue_ln_kernel<T, 3>), grid, dim3(256), 0, st, slabs, slab_elems, S, M, N, ep); break;
    default: hipLaunchKernelGGL((splitk_epilogue_ln_kernel<T, 4>), grid, dim3(256), 0, st, slabs, slab_elems, S, M, N, ep); break;
  }
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

// how many K ranges for a product with `tiles` output tiles of 128 x 128 and `nt` K tiles; 1 = do not split
int splitk_choice(int64_t tiles, int nt, bool ln_fused) {
  static const int off = getenv("IMT_GEMM_NO_SMALL_SPLITK") ? 1 : 0;  // tuning / tests
  // with the LayerNorm in the epilogue launch the split costs no extra launch (it replaces the LayerNorm's), so a
  // shorter K pays already when the product has only a handful of tiles (decoding: 320 rows)
  static const int ln_min_nt = getenv("IMT_GEMM_SPLITK_LN_MIN_NT") ? atoi(getenv("IMT_GEMM_SPLITK_LN_MIN_NT")) : 8;
  const bool short_ok = ln_fused && tiles <= 32 && nt >= ln_min_nt;
  if (off || tiles > 128 || (nt < 16 && !short_ok)) return 1;
  int s = (int)(256 / tiles);
  if (s > 8) s = 8;
  const int per_min = (nt < 16) ? 2 : 4;  // K tiles per range: below that the ramp of a range costs more than it saves
  if (s > nt / per_min) s = nt / per_min;
  if (s < 2) return 1;
  const int per = (nt + s - 1) / s;
  return (nt + per - 1) / per;  // every range non-empty
}

}  // namespace

extern "C" int64_t imt_gemm_splitk_ws_bytes(void) { return (int64_t)256 * BM * BN * 4 + 4096; }

// aux_mode IMT_AUX_SPLITK_WS: a product with few output tiles and a very long K (dX through the vocabulary: 8128 x 512 x
// 30000 has 64 tiles of 256 x 256) runs as split_k K-ranges of 256-tile workgroups, each into its own fp32 slab of the
// caller's workspace `aux` (split_k * M * N floats), followed by one reduce launch -- no atomics, fixed summation order.
static int gemm_splitk_slabs(const imt_gemm_args* a, void* stream) {
  const int bk = (a->dtype == IMT_BF16) ? 64 : 32, es = (a->dtype == IMT_BF16) ? 2 : 4;
  IMT_CHECK_ARG(a->layout != IMT_TN && a->aux && a->split_k >= 2 && a->split_k <= 16, "imt_gemm: split-K slab mode needs NT/NN, a workspace and 2..16 splits");
  IMT_CHECK_ARG(!a->bias && !a->resid && a->dropout_p == 0.f && !a->a_colsum && a->N % 4 == 0 && a->K >= 2 * bk * a->split_k,
                "imt_gemm: split-K slab mode: plain epilogue, N %% 4 == 0, at least two K tiles per split");
  IMT_CHECK_ARG(a->c_dtype == a->dtype || a->c_dtype == IMT_F32, "imt_gemm: c_dtype must be f32 or dtype");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  imt_gemm_args body = *a;
  body.aux = nullptr; body.aux_mode = IMT_AUX_NONE;
  int accumulate = a->accumulate;
  if (a->K % bk != 0) {  // ragged tail first, as its own small product straight into C
    const int kb = (a->K / bk) * bk;
    imt_gemm_args tail = body;
    tail.split_k = 1; tail.K = a->K - kb;
    tail.A = reinterpret_cast<const char*>(a->A) + (int64_t)kb * es;                                   // NT / NN: A is [M][K]
    tail.B = reinterpret_cast<const char*>(a->B) + (a->layout == IMT_NT ? (int64_t)kb : (int64_t)kb * a->ldb) * es;
    const int rc = imt_gemm(&tail, stream);
    if (rc != IMT_OK) return rc;
    body.K = kb; accumulate = 1;
  }
  const int64_t a_rows = body.M, b_rows = (body.layout == IMT_NT) ? body.N : body.K;
  IMT_CHECK_ARG(a_rows * body.lda * es < (1ll << 31) && b_rows * body.ldb * es < (1ll << 31), "imt_gemm: split-K slab mode: operand too large for 32-bit offsets");
  EpiParams ep;
  memset(&ep, 0, sizeof(ep));
  ep.C = a->aux; ep.ldc = body.N; ep.c_f32 = 1; ep.accumulate = 0;
  ep.alpha = body.alpha; ep.alpha_dev = body.alpha_dev; ep.inv_keep = 1.0f;
  ep.slab_elems = (int64_t)body.M * body.N;
  int rc = (body.dtype == IMT_F32) ? dispatch<float>(&body, ep, body.split_k, body.K, 6, st) : dispatch<bf16_t>(&body, ep, body.split_k, body.K, 6, st);
  if (rc != IMT_OK) return rc;
  const int64_t groups = (int64_t)body.M * (body.N / 4);
  ImtProfScope prof("gemm_splitk_reduce", 0.0, (double)body.M * body.N * (4.0 * body.split_k + 2.0 * es), st);
  const float* slabs = reinterpret_cast<const float*>(a->aux);
  if (a->c_dtype == IMT_F32)
    hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(imt_cdiv(groups, 256)), dim3(256), 0, st, slabs, ep.slab_elems, body.split_k,
                       reinterpret_cast<float*>(a->C), a->ldc, body.M, body.N, accumulate);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, dim3(imt_cdiv(groups, 256)), dim3(256), 0, st, slabs, ep.slab_elems, body.split_k,
                       reinterpret_cast<bf16_t*>(a->C), a->ldc, body.M, body.N, accumulate);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

// one-tile-per-CU GEMMs on the three-workgroups-per-CU kernel (data-parallel runs): imt_set_gemm_share_cus, overridden by the
// environment variable IMT_GEMM_SHARE_CUS=0|1
static int g_share_cus = 0;
static bool share_cus_policy() {
  static const char* env = getenv("IMT_GEMM_SHARE_CUS");
  if (env) return atoi(env) != 0 || env[0] == '\0';
  return g_share_cus != 0;
}
bool imt_gemm_ln_ticket_enabled() { return IMT_LN_TICKET != 0; }
extern "C" int imt_set_gemm_share_cus(int share_cus) {
  const int prev = g_share_cus;
  g_share_cus = share_cus ? 1 : 0;
  return prev;
}

extern "C" int imt_gemm(const imt_gemm_args* a, void* stream) {
  IMT_CHECK_ARG(a != nullptr, "imt_gemm: null args");
  IMT_CHECK_ARG(a->dtype == IMT_F32 || a->dtype == IMT_BF16, "imt_gemm: bad dtype %d", a->dtype);
  IMT_CHECK_ARG(a->M >= 0 && a->N >= 0 && a->K >= 0, "imt_gemm: negative dims");
  if (a->M == 0 || a->N == 0) return IMT_OK;
  IMT_CHECK_ARG(a->A && a->B && a->C, "imt_gemm: null operand");
  const int al = (a->dtype == IMT_BF16) ? 8 : 4;
  const int es = (a->dtype == IMT_BF16) ? 2 : 4;
  IMT_CHECK_ARG(a->lda % al == 0 && a->ldb % al == 0, "imt_gemm: lda/ldb must be multiples of %d (16-B rows)", al);
  IMT_CHECK_ARG(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "imt_gemm: A/B must be 16-B aligned");
  // contiguous extents of the vector loads must be chunk multiples
  const int a_inner = (a->layout == IMT_TN) ? a->M : a->K;
  const int b_inner = (a->layout == IMT_NT) ? a->K : a->N;
  // A contiguous extent that is not a whole number of 16-byte chunks (a vocabulary of 193 entries ...) is accepted when
  // the leading dimension covers the rounded-up extent (the last chunk of a row then stays inside the row's storage)
  // and what the overhang reads cannot reach the result: output rows/columns past M/N are masked by the epilogue, and
  // the K overhang of the K-contiguous A of NN meets zero-filled B rows.  Both operands of NT share the K overhang,
  // so NT keeps the strict rule.
  const auto extent_ok = [al](int inner, int64_t ld, bool strict) {
    return inner % al == 0 || (!strict && ld >= (int64_t)((inner + al - 1) / al) * al);
  };
  IMT_CHECK_ARG(extent_ok(a_inner, a->lda, a->layout == IMT_NT) && extent_ok(b_inner, a->ldb, a->layout == IMT_NT),
                "imt_gemm: contiguous extents (%d,%d) must be multiples of %d (or, for NN/TN, padded by the leading dimension)",
                a_inner, b_inner, al);
  IMT_CHECK_ARG(a->ldc % 4 == 0, "imt_gemm: ldc must be a multiple of 4");
  const int c_f32 = (a->c_dtype == IMT_F32);
  IMT_CHECK_ARG(c_f32 || a->c_dtype == a->dtype, "imt_gemm: c_dtype must be f32 or dtype");
  IMT_CHECK_ARG(!a->a_colsum || a->layout == IMT_TN, "imt_gemm: a_colsum is a TN (weight-gradient) option");
  if (a->aux_mode == IMT_AUX_SPLITK_WS) return gemm_splitk_slabs(a, stream);
  int splits = a->split_k > 1 ? a->split_k : 1;
  const int bk = (a->dtype == IMT_BF16) ? 64 : 32;
  int kps = a->K;
  if (splits > 1) {
    IMT_CHECK_ARG(c_f32 && !a->bias && !a->resid && a->aux_mode == IMT_AUX_NONE && a->dropout_p == 0.f,
                  "imt_gemm: split_k supports only fp32 atomic accumulation without epilogue");
    kps = imt_cdiv(imt_cdiv(a->K, splits), bk) * bk;
    splits = imt_cdiv(a->K, kps);
  }
  if (a->aux_mode != IMT_AUX_NONE) IMT_CHECK_ARG(a->aux != nullptr, "imt_gemm: aux_mode needs aux");
  // LDS-DMA pipeline: whole K tiles only (no zero-fill needed anywhere), 32-bit buffer offsets
  const int64_t a_rows = (a->layout == IMT_TN) ? a->K : a->M, b_rows = (a->layout == IMT_NT) ? a->N : a->K;
  // (TN: K-strided operands, a ragged last K tile is zero-filled by the descriptors' range check -- any K)
  const bool pipe_ok = (a->K % bk == 0 || a->layout == IMT_TN) && (a->K / bk >= 2) && (a_rows * a->lda * es < (1ll << 31)) &&
                       (b_rows * a->ldb * es < (1ll << 31));
  // Long K that is not a whole number of tiles (the vocabulary dimension: dX = dlogits[.,30000] W): the LDS-DMA
  // kernels need whole K tiles, so the ragged tail goes first as its own small product and the whole-tile body is
  // accumulated on top by the persistent kernel (477 -> ~700 TFLOP/s on 8128x512x30000).  Only for epilogues that are
  // linear in the product (bias / residual ride on the tail call).
  if (a->force_general == 0 && splits == 1 && a->layout != IMT_TN && a->K % bk != 0 && a->K / bk >= 16 && a->aux_mode == IMT_AUX_NONE &&
      a->dropout_p == 0.f && !a->a_colsum) {
    const int kb = (a->K / bk) * bk, kt = a->K - kb;
    imt_gemm_args tail = *a, body = *a;
    const int64_t a_off = (a->layout == IMT_TN) ? (int64_t)kb * a->lda : kb;
    const int64_t b_off = (a->layout == IMT_NT) ? kb : (int64_t)kb * a->ldb;
    tail.K = kt;
    tail.A = reinterpret_cast<const char*>(a->A) + a_off * es;
    tail.B = reinterpret_cast<const char*>(a->B) + b_off * es;
    body.K = kb; body.accumulate = 1; body.bias = nullptr; body.resid = nullptr;
    tail.ln_out = nullptr; body.ln_out = nullptr;  // the LayerNorm (if any) follows the complete product
    if (a->ln_out) {
      IMT_CHECK_ARG(a->ln_gamma && a->ln_beta, "imt_gemm: ln_out needs ln_gamma and ln_beta");
      IMT_CHECK_ARG(a->c_dtype == a->dtype && !a->accumulate, "imt_gemm: ln_out needs C of the compute type, no accumulate");
      IMT_CHECK_ARG(a->ldc == a->N && a->ld_ln == a->N, "imt_gemm: ln_out behind a GEMM that cannot normalise in-launch needs contiguous C and ln_out");
    }
    int rc = imt_gemm(&tail, stream);
    if (rc != IMT_OK) return rc;
    rc = imt_gemm(&body, stream);
    if (rc != IMT_OK || !a->ln_out) return rc;
    return imt_layernorm_fwd(a->dtype, a->C, a->ln_gamma, a->ln_beta, a->ln_out, a->ln_mean, a->ln_rstd, a->M, a->N, a->ln_eps, 0.f, 0, stream);
  }
  // few output tiles, long K, a workspace for partial sums: K ranges on the persistent kernel + one epilogue launch
  if (a->force_general == 0 && a->splitk_ws && splits == 1 && a->layout != IMT_TN && pipe_ok && a->N % 4 == 0 && !a->a_colsum) {
    const int64_t tiles = (int64_t)imt_cdiv(a->M, BM) * imt_cdiv(a->N, BN);
    const bool ln_fused = a->ln_out && a->aux_mode == IMT_AUX_NONE && a->N <= 1024 && !c_f32;
    const int S = splitk_choice(tiles, a->K / bk, ln_fused);
    const int64_t rows_pad = (int64_t)imt_cdiv(a->M, BM) * BM;  // edge tiles store only rows < M, slabs are addressed [m][n] with ld N
    if (S > 1 && (int64_t)S * a->M * a->N * 4 <= a->splitk_ws_bytes && rows_pad > 0 && ((uintptr_t)a->splitk_ws & 15) == 0) {
      if (a->aux_mode != IMT_AUX_NONE) IMT_CHECK_ARG(a->aux != nullptr, "imt_gemm: aux_mode needs aux");
      if (a->ln_out) {
        IMT_CHECK_ARG(a->ln_gamma && a->ln_beta, "imt_gemm: ln_out needs ln_gamma and ln_beta");
        IMT_CHECK_ARG(a->c_dtype == a->dtype && !a->accumulate && a->ldc == a->N && a->ld_ln == a->N, "imt_gemm: ln_out needs contiguous C of the compute type");
      }
      hipStream_t st = reinterpret_cast<hipStream_t>(stream);
      EpiParams slab;
      slab.C = a->splitk_ws; slab.ldc = a->N; slab.c_f32 = 1; slab.accumulate = 0;
      slab.bias = nullptr; slab.resid = nullptr; slab.ldr = 0; slab.aux = nullptr; slab.ldaux = 0; slab.aux_mode = IMT_AUX_NONE;
      slab.atomic = 0; slab.alpha = 1.0f; slab.alpha_dev = nullptr; slab.inv_keep = 1.0f; slab.drop_thresh = 0; slab.seed = 0;
      slab.a_colsum = nullptr; slab.dbg = 0; slab.trace = nullptr;
      slab.slab_elems = (int64_t)a->M * a->N; slab.splits = S;
      int rc = (a->dtype == IMT_F32) ? dispatch<float>(a, slab, 1, a->K, 5, st) : dispatch<bf16_t>(a, slab, 1, a->K, 5, st);
      if (rc != IMT_OK) return rc;
      EpiParams ep;
      ep.C = a->C; ep.ldc = a->ldc; ep.c_f32 = c_f32; ep.accumulate = a->accumulate;
      ep.bias = a->bias; ep.resid = a->resid; ep.ldr = a->ldr;
      ep.aux = a->aux; ep.ldaux = a->ldaux; ep.aux_mode = a->aux_mode; ep.atomic = 0;
      ep.alpha = a->alpha; ep.alpha_dev = a->alpha_dev;
      ep.drop_thresh = dropout_thresh(a->dropout_p);
      ep.inv_keep = a->dropout_p > 0.f ? 1.0f / (1.0f - a->dropout_p) : 1.0f;
      ep.seed = a->dropout_seed; ep.a_colsum = nullptr; ep.dbg = 0; ep.trace = nullptr; ep.slab_elems = 0;
      if (ln_fused) {
        // dense + bias + dropout + residual + LayerNorm: the epilogue launch normalises its rows itself
        ep.ln_gamma = a->ln_gamma; ep.ln_beta = a->ln_beta; ep.ln_out = a->ln_out; ep.ld_ln = a->ld_ln;
        ep.ln_mean = a->ln_mean; ep.ln_rstd = a->ln_rstd; ep.ln_eps = a->ln_eps;
        ImtProfScope prof("gemm_splitk_epilogue_ln", 0.0, (double)a->M * a->N * (4.0 * S + 3.0 * es), st);
        const float* slabs = reinterpret_cast<const float*>(a->splitk_ws);
        return (a->dtype == IMT_F32) ? launch_splitk_epilogue_ln<float>(slabs, slab.slab_elems, S, a->M, a->N, ep, st)
                                     : launch_splitk_epilogue_ln<bf16_t>(slabs, slab.slab_elems, S, a->M, a->N, ep, st);
      }
      {
        const int64_t groups = (int64_t)a->M * (a->N / 4);
        ImtProfScope prof("gemm_splitk_epilogue", 0.0, (double)a->M * a->N * (4.0 * S + 2.0 * es), st);
        const float* slabs = reinterpret_cast<const float*>(a->splitk_ws);
        if (a->dtype == IMT_F32)
          hipLaunchKernelGGL(splitk_epilogue_kernel<float>, dim3(imt_cdiv(groups, 256)), dim3(256), 0, st, slabs, slab.slab_elems, S, a->M, a->N, ep);
        else
          hipLaunchKernelGGL(splitk_epilogue_kernel<bf16_t>, dim3(imt_cdiv(groups, 256)), dim3(256), 0, st, slabs, slab.slab_elems, S, a->M, a->N, ep);
        IMT_CHECK_LAUNCH();
      }
      if (!a->ln_out) return IMT_OK;
      return imt_layernorm_fwd(a->dtype, a->C, a->ln_gamma, a->ln_beta, a->ln_out, a->ln_mean, a->ln_rstd, a->M, a->N, a->ln_eps, 0.f, 0, stream);
    }
  }
  const int64_t nblocks = (int64_t)imt_cdiv(a->M, BM) * imt_cdiv(a->N, BN) * splits;
  // kernel variant: 1 = register-staged double buffer (2 blocks/CU), 2 = LDS-DMA 3-stage ring (1 block/CU),
  // 3 = single buffer + register prefetch (4 blocks/CU).  a->force_general carries a variant code for tests/tuning.
  int variant = a->force_general % 100;
  static const int env_dbg = getenv("IMT_GEMM_DBG") ? atoi(getenv("IMT_GEMM_DBG")) : 0;  // tuning only
  const int dbg = a->force_general / 100 | env_dbg;
  // measured on MI355X (profiles/r01_v3_gemm_shapes.txt): about one wave of blocks -> the LDS-DMA ring (its long
  // steady state wins when K >= 1024, a tie otherwise); larger grids -> three single-buffer blocks per CU.
  // measured on MI355X (profiles/r01_v5_gemm_shapes.txt): the persistent wave-specialised kernel wins whenever a CU
  // gets about one output tile or K is long (740 vs 480 TFLOP/s at 8192x512x2048); grids of many short-K tiles are
  // still better served by three single-buffer blocks per CU overlapping each other's epilogues.
  if (variant == 0) {
    const int64_t tiles = (int64_t)imt_cdiv(a->M, BM) * imt_cdiv(a->N, BN);
    // (weight gradients with many output tiles and a short K -- few tokens: captioning, decoding-sized batches -- also run
    // best on the persistent kernel: 30000 x 512 x 992 TN 64 against 92 us on the register-staged one, tools/tn_small_k.py)
    if (pipe_ok && splits == 1 && (tiles <= 256 || a->K >= 1024 || a->layout == IMT_TN)) variant = 5;
    else variant = (a->layout == IMT_TN) ? 1 : 3;
    // 256 x 256 tiles (profiles/r01_gemm_xl_study.txt): several rounds of short-K tiles (vocabulary projection 797 vs 681
    // TFLOP/s, batched cross K/V), or one round whose epilogue touches a second matrix (GELU / GELU' / residual: 500 vs 430)
    const int64_t tiles256 = (int64_t)imt_cdiv(a->M, 256) * imt_cdiv(a->N, 256);
    static const bool no_xl = getenv("IMT_GEMM_NO_XL") != nullptr;
    static const bool no_xl_gelu = getenv("IMT_GEMM_NO_XL_GELU") != nullptr;  // tuning only
    if (!no_xl && pipe_ok && splits == 1 && a->layout != IMT_TN && !a->a_colsum && a->K < 1024 &&
        (tiles256 >= 512 || (tiles256 >= 224 && ((a->aux_mode != IMT_AUX_NONE && !(a->aux_mode == IMT_AUX_GELU_FWD && no_xl_gelu)) || a->resid))))
      variant = 6;
    // a weight gradient with about one 256-tile per CU and a long K (the vocabulary projection's dW: 30000 x 512 x 8128)
    static const bool no_xl_tn = getenv("IMT_GEMM_NO_XL_TN") != nullptr;
    if (!no_xl && !no_xl_tn && pipe_ok && splits == 1 && a->layout == IMT_TN && tiles256 >= 224 && tiles256 <= 256 && a->K >= 2048) variant = 6;
    // data-parallel knob (off by default): when a collective's kernels hold some CUs, a persistent
    // launch of exactly one tile per CU needs a full second round; three small workgroups per CU degrade gracefully
    if (share_cus_policy() && variant == 5 && a->layout != IMT_TN && tiles > 192 && tiles <= 256 && a->K < 1024) variant = 3;
  }
  if (variant == 6 && (!pipe_ok || splits > 1 || (a->a_colsum && a->layout != IMT_TN))) variant = 3;
  if ((variant == 2 || variant == 4) && !pipe_ok) variant = 1;
  if (variant == 5 && (!pipe_ok || splits > 1)) variant = 3;
  EpiParams ep;
  ep.C = a->C; ep.ldc = a->ldc; ep.c_f32 = c_f32; ep.accumulate = a->accumulate;
  ep.bias = a->bias; ep.resid = a->resid; ep.ldr = a->ldr;
  ep.aux = a->aux; ep.ldaux = a->ldaux; ep.aux_mode = a->aux_mode;
  ep.atomic = (splits > 1);
  ep.alpha = a->alpha; ep.alpha_dev = a->alpha_dev;
  ep.drop_thresh = dropout_thresh(a->dropout_p);
  ep.inv_keep = a->dropout_p > 0.f ? 1.0f / (1.0f - a->dropout_p) : 1.0f;
  ep.seed = a->dropout_seed;
  ep.a_colsum = a->a_colsum;
  ep.dbg = dbg;
  ep.trace = nullptr;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // LayerNorm of the finished rows: in-launch (ln_rowblock_tail) when this is a one-tile-per-workgroup launch of the
  // persistent kernel over whole 128-column tiles; otherwise a LayerNorm launch behind the GEMM
  bool ln_in_launch = false;
  if (a->ln_out) {
    IMT_CHECK_ARG(a->ln_gamma && a->ln_beta, "imt_gemm: ln_out needs ln_gamma and ln_beta");
    IMT_CHECK_ARG(a->c_dtype == a->dtype && splits == 1 && !a->accumulate, "imt_gemm: ln_out needs C of the compute type, no split-K, no accumulate");
    static const bool no_ticket = getenv("IMT_GEMM_LN_TICKET") && atoi(getenv("IMT_GEMM_LN_TICKET")) == 0;  // tuning / tests
    const int64_t tiles = (int64_t)imt_cdiv(a->M, BM) * imt_cdiv(a->N, BN);
    ln_in_launch = IMT_LN_TICKET && !no_ticket && variant == 5 && tiles <= 256 && a->N % BN == 0 && a->N <= 1024 && a->ln_tickets != nullptr &&
                   ((int64_t)(a->M - 1) * a->ldc + a->N) * es < (1ll << 31);
    if (ln_in_launch) {
      ep.ln_gamma = a->ln_gamma; ep.ln_beta = a->ln_beta; ep.ln_out = a->ln_out; ep.ld_ln = a->ld_ln;
      ep.ln_mean = a->ln_mean; ep.ln_rstd = a->ln_rstd; ep.ln_tickets = a->ln_tickets; ep.ln_eps = a->ln_eps;
    } else {
      IMT_CHECK_ARG(a->ldc == a->N && a->ld_ln == a->N, "imt_gemm: ln_out behind a GEMM that cannot normalise in-launch needs contiguous C and ln_out");
    }
  }
  const int rc = (a->dtype == IMT_F32) ? dispatch<float>(a, ep, splits, kps, variant, st) : dispatch<bf16_t>(a, ep, splits, kps, variant, st);
  if (rc != IMT_OK || !a->ln_out || ln_in_launch) return rc;
  return imt_layernorm_fwd(a->dtype, a->C, a->ln_gamma, a->ln_beta, a->ln_out, a->ln_mean, a->ln_rstd, a->M, a->N, a->ln_eps, 0.f, 0, stream);
}

extern "C" int imt_gemm_grouped_tn(const imt_gemm_args* list, int count, void* stream) {
  IMT_CHECK_ARG(list && count >= 0, "imt_gemm_grouped_tn: bad args");
  if (count == 0) return IMT_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  bool ok = count <= MAX_GROUP;
  const int dtype = list[0].dtype;
  const int bk = (dtype == IMT_BF16) ? 64 : 32, es = (dtype == IMT_BF16) ? 2 : 4, al = (dtype == IMT_BF16) ? 8 : 4;
  int tiles = 0;
  for (int i = 0; i < count && ok; ++i) {
    const imt_gemm_args& a = list[i];
    ok = a.dtype == dtype && a.layout == IMT_TN && a.c_dtype == IMT_F32 && !a.bias && !a.resid && a.aux_mode == IMT_AUX_NONE &&
         a.dropout_p == 0.f && !a.alpha_dev && a.alpha == list[0].alpha && a.A && a.B && a.C && a.M > 0 && a.N > 0 &&
         a.K / bk >= 2 && a.lda % al == 0 && a.ldb % al == 0 && a.M % al == 0 && a.N % al == 0 && a.ldc % 4 == 0 &&
         (((uintptr_t)a.A | (uintptr_t)a.B) & 15) == 0 && (int64_t)a.K * a.lda * es < (1ll << 31) &&
         (int64_t)a.K * a.ldb * es < (1ll << 31);
    tiles += imt_cdiv(a.M, BM) * imt_cdiv(a.N, BN);
  }
  if (!ok || tiles > 640) {
    // not groupable (ragged K, too many problems, ...): individual launches with their own split-K choice
    for (int i = 0; i < count; ++i) {
      int rc = imt_gemm(&list[i], stream);
      if (rc != IMT_OK) return rc;
    }
    return IMT_OK;
  }
  GroupArgs g;
  memset(&g, 0, sizeof(g));
  g.count = count; g.alpha = list[0].alpha;
  static const bool plain_order = getenv("IMT_GROUPED_PLAIN_ORDER") != nullptr;  // tuning only
  g.alpha_pad_order = plain_order ? 1 : 0;
  int start = 0;
  double flops = 0, bytes = 0;
  for (int i = 0; i < count; ++i) {
    const imt_gemm_args& a = list[i];
    GroupProblem& P = g.p[i];
    P.A = a.A; P.B = a.B; P.C = reinterpret_cast<float*>(a.C); P.a_colsum = a.a_colsum;
    P.lda = a.lda; P.ldb = a.ldb; P.ldc = a.ldc;
    P.a_bytes = view_bytes(a.K, a.lda, a.M, es); P.b_bytes = view_bytes(a.K, a.ldb, a.N, es);
    P.M = a.M; P.N = a.N; P.K = a.K; P.tile_start = start;
    start += imt_cdiv(a.M, BM) * imt_cdiv(a.N, BN);
    flops += 2.0 * a.M * a.N * a.K;
    bytes += ((double)a.M + a.N) * a.K * es + 4.0 * a.M * a.N;
  }
  g.total_tiles = start;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_grouped_tn_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, GROUP_NST * STAGE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_grouped_tn_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, GROUP_NST * STAGE_BYTES);
    attr_set = true;
  }
  ImtProfScope prof(dtype == IMT_BF16 ? "gemm_bf16_tn_grouped" : "gemm_f32_tn_grouped", flops, bytes, st);
  if (dtype == IMT_F32)
    hipLaunchKernelGGL(gemm_grouped_tn_kernel<float>, dim3(start), dim3(GROUP_THREADS), GROUP_NST * STAGE_BYTES, st, g);
  else
    hipLaunchKernelGGL(gemm_grouped_tn_kernel<bf16_t>, dim3(start), dim3(GROUP_THREADS), GROUP_NST * STAGE_BYTES, st, g);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}
