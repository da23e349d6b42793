// One decoding step of a whole cross-attending decoder stack in ONE launch (bf16, hidden size 512 or 768 in heads of 64; src/seq_gen.py:164-194 calls
// the decoder once per step on B x beam hypothesis rows).  The launch-per-operator path (model.hip: imt_decode_step) spends a
// step's 0.66 ms in 67 launches of ~8 us each whose work is a few hundred rows: the time is launch boundaries, cold weight
// fetches and write-backs.  Here 256 workgroups (one per CU, all resident) walk the layers together:
//   per layer   P1 q|k|v of the new position -> cache      P2 self attention over the cache      P3 output projection + residual
//               P4 LayerNorm -> cross query                 P5 cross attention over the encoder    P6 output projection + residual
//               P7 LayerNorm -> FFN-up + GELU               P8 FFN-down + residual                 (LayerNorm of P8: next layer's P1)
// separated by grid-wide barriers (tools/probe_gridbar.hip: 2.1 us with per-XCD arrival counters) instead of launches.  A product
// is cut into items of 32 (64) rows x 32|64 columns (160-480 items); the weight tile of a workgroup's NEXT item is requested
// into LDS BEFORE the barrier (weights depend on nobody), the activation rows after it.  A LayerNorm is computed by the consumer
// while it stages its A operand (fp32 pre-LayerNorm rows -> registers -> normalised bf16 tile in LDS; the items of column
// tile 0 also store the normalised rows for the residual two phases later).  Hand-offs follow the micro-architecture guide's
// Guideline 16: write-through (sc1) stores, every wave drains, barrier; sc1 loads; every hand-off buffer is written ONCE per
// launch (per-layer buffers), so no workgroup can hold a stale copy.  Every spin is bounded; a timeout sets a sticky status word
// that makes every workgroup leave (imt_decode_check reports it).
#include "mma.hpp"
#include "decode_attn.hpp"
#include "decode_fused.hpp"
#include <stdlib.h>

namespace {

typedef bf16_t T;
typedef Frag<T>::type frag_t;
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr int FTHREADS = 1024, FWAVES = FTHREADS / 64;
// 16 waves per workgroup: the attention phases are one wave per (hypothesis, head) -- 2560 of them at 64 x 5 x 8 -- and latency-bound, so they
// want every wave slot of the chip (4096); the products use waves 0-3 (0-7) for the MFMAs and all 16 for staging.
constexpr int FUSED_UNR = 5;   // key groups in flight per attention wave (6 and 8 spill at the 128 registers of 16 waves per CU)

// Geometry per hidden size D (512: BASELINE configs; 768: the reference's default --embed).  A product with K = D holds its whole K in
// LDS (one tile per operand); FFN-down streams K = ff in chunks of KC2 through two slots per operand.
template <int D> struct FCfg {
  static_assert(D == 512 || D == 768, "hidden sizes the one-launch step is built for");
  static constexpr int NJ = (D + 511) / 512;            // 8-column groups of a row per lane: columns 8 lane + 512 j
  static constexpr int KC2 = D == 768 ? 384 : 512;      // K per chunk of the streamed product
  static constexpr int BRT_UP = D == 512 ? 64 : 32;     // rows per FFN-up item (64 rows x D need 64 KiB at 512, 96 at 768)
  static constexpr int W_REGION = 64 * D * 2;           // one 64-column weight tile of depth D (>= two 32-column chunks of KC2)
  static constexpr int A_REGION = BRT_UP * D * 2;       // the item's rows at depth D (>= two 32-row chunks of KC2)
  static constexpr int LDS = W_REGION + A_REGION;       // 128 KiB / 144 KiB
  static_assert(2 * 32 * KC2 * 2 <= W_REGION && 2 * 32 * KC2 * 2 <= A_REGION, "");
};

IMT_DEVICE unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
IMT_DEVICE void st_rlx(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
IMT_DEVICE unsigned add_rlx(unsigned* p, unsigned v) { return __hip_atomic_fetch_add((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
IMT_DEVICE f32x4 ld_sc1_bf16x4(const T* p) {  // 8 bytes written by another workgroup of this launch
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bf16x4 b = __builtin_bit_cast(bf16x4, v);
  return f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
}
IMT_DEVICE void wait_vmcnt(int n) {   // n: wave-uniform, 0 .. 8
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  }
}

struct GridBar {
  unsigned* w;   // [0] global count, [32 (1 + x)] arrivals of XCD x, [IMT_FUSED_BAR_WORDS - 1] status
  int round;
  int* give_up;  // LDS
  unsigned long long* trace;   // tuning (IMT_DECODE_TRACE=1): [workgroup][2 x barriers] stamps = arrival at / departure from each barrier
  // every wave's stores are out (write-through), then one thread arrives; false: the launch is being abandoned
  IMT_DEVICE bool sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (trace && threadIdx.x == 0) trace[(int64_t)blockIdx.x * 128 + 2 * round] = __builtin_amdgcn_s_memrealtime();
    ++round;
    if (threadIdx.x == 0) {
      const int xcd = blockIdx.x & 7, nx = gridDim.x >> 3;
      unsigned* status = w + IMT_FUSED_BAR_WORDS - 1;
      const unsigned prev = add_rlx(w + 32 * (1 + xcd), 1u);
      if (prev + 1u == (unsigned)round * (unsigned)nx) add_rlx(w, 1u);
      bool ok = false;
      for (unsigned s = 0; s < 8000000u; ++s) {
        if (ld_rlx(w) >= (unsigned)round * 8u) { ok = true; break; }
        if ((s & 127u) == 127u && ld_rlx(status) != 0) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) { st_rlx(status, 0x1000u + (unsigned)round); *give_up = 1; }
    }
    __syncthreads();
    if (trace && threadIdx.x == 0) trace[(int64_t)blockIdx.x * 128 + 2 * round - 1] = __builtin_amdgcn_s_memrealtime();
    return *give_up == 0;
  }
};

IMT_DEVICE __amdgpu_buffer_rsrc_t rsrc_of(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// [NROWS rows][KC k] of a K-contiguous bf16 operand -> LDS as KC / 64 sub-tiles [NROWS][64 k] (128-byte rows, the library's swizzle);
// rows / bytes past the descriptor's range read zeros.  One-KiB pieces dealt round-robin to the 16 waves.
template <int NROWS, int KC> IMT_DEVICE int dma_loads_of_wave(int wave) { return ((NROWS / 8) * (KC / 64) - wave + FWAVES - 1) / FWAVES; }
template <int NROWS, int KC, bool SC1>
IMT_DEVICE void dma_rows(__amdgpu_buffer_rsrc_t rs, int64_t ld, int row0, int k0, char* dst) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int PPT = NROWS / 8, PIECES = PPT * (KC / 64);
#pragma unroll
  for (int i = 0; i < (PIECES + FWAVES - 1) / FWAVES; ++i) {
    const int p = wave + FWAVES * i;
    if (PIECES % FWAVES == 0 || p < PIECES) {   // (wave-uniform)
      const int kt = p / PPT, pi = p % PPT;
      const int row = 8 * pi + (lane >> 3), c = (lane & 7) ^ swz<128>(row);
      const int voff = (int)((((int64_t)(row0 + row)) * ld + k0 + kt * 64 + c * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + kt * (NROWS * 128) + pi * 1024), 16, voff,
                                               0, 0, SC1 ? 16 : 0);
    }
  }
}

// A row of D columns on a wave: lane l holds the 8-column groups 8 l + 512 j (j < NJ; at D = 768 the second group exists for l < 32)
template <int D> struct RowRaw { u32x4 a[FCfg<D>::NJ], b[FCfg<D>::NJ]; };
template <int D> IMT_DEVICE bool group_ok(int j) { return 8 * (int)(threadIdx.x & 63) + 512 * j < D; }

// LayerNorm of one fp32 row.  The row is REQUESTED (sc1 loads; rows past the descriptor read zeros) apart from being normalised, so
// that a wave has all its rows in flight before the first reduction
template <int D> IMT_DEVICE RowRaw<D> ln_request(__amdgpu_buffer_rsrc_t rs, int grow) {
  const int lane = threadIdx.x & 63;
  RowRaw<D> r;
#pragma unroll
  for (int j = 0; j < FCfg<D>::NJ; ++j) {
    const int off = group_ok<D>(j) ? (grow * D + 8 * lane + 512 * j) * 4 : 0x7ffffff0;   // past every descriptor: zeros
    r.a[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
    r.b[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (group_ok<D>(j) ? 16 : 0), 0, 16);
  }
  return r;
}
template <int D>
IMT_DEVICE void ln_finish(const RowRaw<D>& r, const float (&g)[FCfg<D>::NJ][8], const float (&be)[FCfg<D>::NJ][8], float eps, bf16x8 (&y)[FCfg<D>::NJ]) {
  constexpr int NJ = FCfg<D>::NJ;
  float x[NJ][8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const f32x4 v0 = __builtin_bit_cast(f32x4, r.a[j]), v1 = __builtin_bit_cast(f32x4, r.b[j]);
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[j][e] = v0[e]; x[j][4 + e] = v1[e]; }
#pragma unroll
    for (int e = 0; e < 8; ++e) s += x[j][e];   // (absent groups hold zeros)
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (group_ok<D>(j)) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = x[j][e] - mean; q += t * t; }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) y[j][e] = (bf16_t)((x[j][e] - mean) * rstd * g[j][e] + be[j][e]);
}
template <int D>
IMT_DEVICE void load_gamma_beta(const T* gamma, const T* beta, float (&g)[FCfg<D>::NJ][8], float (&be)[FCfg<D>::NJ][8]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < FCfg<D>::NJ; ++j) {
    const int c = group_ok<D>(j) ? 8 * lane + 512 * j : 0;
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(gamma + c), bv = *reinterpret_cast<const bf16x8*>(beta + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) { g[j][e] = (float)gv[e]; be[j][e] = (float)bv[e]; }
  }
}
IMT_DEVICE void store16_wt(T* p, bf16x8 y) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2 w = __builtin_bit_cast(u64x2, y);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, w[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct GP {
  const T* W; const T* bias; int N, K;              // W [N][K]
  const T* A; int64_t lda;                          // AMODE 0: bf16 rows [R][lda]
  const float* pre; const T *gamma, *beta; T* norm_out;  // AMODE 1: LayerNorm of fp32 rows [R][D]; column tile 0 stores norm_out
  // AMODE 2: LayerNorm of word + position + type embedding rows (gamma / beta / norm_out as above)
  const int64_t *ids, *pos_ids, *type_ids; const T *emb_word, *emb_pos, *emb_type; int vocab, max_pos, n_types;
  T* out; int64_t ldo;                              // EMODE 0 / 2: bf16 (+ GELU)
  float* pre_out; const T* resid; bool resid_sc1;   // EMODE 1: fp32 acc + bias + resid -> pre_out [R][D]
};

// word + position + type embedding of hypothesis row `grow`; indices clamped like imt_embed_ln_fwd; the sum is rounded to bf16 before the
// LayerNorm, as that kernel stores it
template <int D> IMT_DEVICE RowRaw<D> emb_request(const GP& p, int grow, int R) {
  const int lane = threadIdx.x & 63;
  const int rr = min(grow, R - 1);
  int64_t wi = p.ids[rr], pi = p.pos_ids[rr], ti = p.type_ids ? p.type_ids[rr] : 0;
  if (wi < 0 || wi >= p.vocab) wi = 0;
  if (pi < 0 || pi >= p.max_pos) pi = 0;
  if (ti < 0 || ti >= p.n_types) ti = 0;
  RowRaw<D> r;
#pragma unroll
  for (int j = 0; j < FCfg<D>::NJ; ++j) {
    const bool ok = group_ok<D>(j);
    const int col = ok ? 8 * lane + 512 * j : 0;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p.emb_word + wi * D + col);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(p.emb_pos + pi * D + col);
    const bf16x8 c = *reinterpret_cast<const bf16x8*>(p.emb_type + ti * D + col);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = ok ? (float)(bf16_t)(((float)a[e] + (float)b[e]) + (float)c[e]) : 0.f;
    r.a[j] = __builtin_bit_cast(u32x4, f32x4{x[0], x[1], x[2], x[3]});
    r.b[j] = __builtin_bit_cast(u32x4, f32x4{x[4], x[5], x[6], x[7]});
  }
  return r;
}

// BRT rows x (32 NCT) columns on BRT / 8 waves: wave -> row tile wave % (BRT / 16), NCT column tiles from (wave / (BRT / 16)) NCT
template <int BRT, int NCT, int KC>
IMT_DEVICE void mma_chunk(const char* A, const char* W, f32x4 (&acc)[NCT]) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NRT = BRT / 16, BC = 32 * NCT;   // two column groups of NCT tiles whatever BRT is
  const int rt = wave % NRT, ct0 = (wave / NRT) * NCT;
#pragma unroll
  for (int kt = 0; kt < KC / 64; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const frag_t fa = lds_frag_kcontig<T, 128>(A + kt * (BRT * 128), 16 * rt, 4 * s);
#pragma unroll
      for (int j = 0; j < NCT; ++j) mma16(acc[j], lds_frag_kcontig<T, 128>(W + kt * (BC * 128), 16 * (ct0 + j), 4 * s), fa);
    }
}

// the weight tile(s) of this workgroup's first item of a phase, requested before the barrier in front of the phase
template <int BRT, int NCT, int KC>
IMT_DEVICE void prefetch_w(const GP& g, int R, char* smem) {
  constexpr int BC = 32 * NCT;
  const int nct = g.N / BC, nitems = ((R + BRT - 1) / BRT) * nct;
  if ((int)blockIdx.x >= nitems) return;
  const int ct = (int)blockIdx.x % nct;
  const __amdgpu_buffer_rsrc_t rsW = rsrc_of(g.W, (int64_t)g.N * g.K * 2);
  dma_rows<BC, KC, false>(rsW, g.K, ct * BC, 0, smem);
  if (g.K > KC) dma_rows<BC, KC, false>(rsW, g.K, ct * BC, KC, smem + BC * KC * 2);   // (the second slot)
}

// one product: items of BRT rows x BC = 32 NCT columns, K in chunks of KC.  K == KC: one tile per operand (any BRT / NCT that fits the
// regions); K > KC: NCT == 1, BRT == 32, two slots per operand.
template <int D, int BRT, int NCT, int AMODE, int EMODE, int KC>
IMT_DEVICE void gemm_phase(const GP& g, int R, float eps, char* smem, bool w_ready) {
  typedef FCfg<D> C;
  constexpr int BC = 32 * NCT, NJ = C::NJ;
  constexpr int W_SLOT = BC * KC * 2, A_SLOT = BRT * KC * 2, MW = BRT / 8, NRT = BRT / 16, RPW = BRT / FWAVES;
  static_assert(W_SLOT <= C::W_REGION && A_SLOT <= C::A_REGION, "operand tile does not fit its LDS region");
  static_assert(AMODE == 0 || KC == D, "a LayerNorm-staged operand holds whole rows");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nct = g.N / BC, nitems = ((R + BRT - 1) / BRT) * nct, nch = g.K / KC;
  const __amdgpu_buffer_rsrc_t rsW = rsrc_of(g.W, (int64_t)g.N * g.K * 2);
  const __amdgpu_buffer_rsrc_t rsA = AMODE == 0 ? rsrc_of(g.A, (int64_t)R * g.lda * 2) : rsrc_of(AMODE == 1 ? (const void*)g.pre : (const void*)g.W, (int64_t)R * D * 4);
  const int nA = dma_loads_of_wave<BRT, KC>(wave), nW = dma_loads_of_wave<BC, KC>(wave);
  char* Wreg = smem;
  char* Areg = smem + C::W_REGION;
  bool first = true;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int rb = item / nct, ct = item % nct, row0 = rb * BRT, col0 = ct * BC;
    const bool have_w = first && w_ready;   // chunk 0 (and 1) of the weights arrived under the barrier
    if (!first) __syncthreads();  // the previous item's LDS reads are done
    if (!have_w) dma_rows<BC, KC, false>(rsW, g.K, col0, 0, Wreg);
    if (AMODE == 0) {
      dma_rows<BRT, KC, true>(rsA, g.lda, row0, 0, Areg);
    } else {
      float gm[NJ][8], be[NJ][8];
      load_gamma_beta<D>(g.gamma, g.beta, gm, be);
      RowRaw<D> raw[RPW];
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr)
        raw[rr] = AMODE == 2 ? emb_request<D>(g, row0 + wave * RPW + rr, R) : ln_request<D>(rsA, row0 + wave * RPW + rr);
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int row = wave * RPW + rr;
        bf16x8 y[NJ];
        ln_finish<D>(raw[rr], gm, be, eps, y);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          if (group_ok<D>(j)) {
            *reinterpret_cast<bf16x8*>(Areg + ((lane >> 3) + 8 * j) * (BRT * 128) + tile_off<128>(row, lane & 7)) = y[j];
            if (ct == 0 && g.norm_out && row0 + row < R) store16_wt(g.norm_out + (int64_t)(row0 + row) * D + 8 * lane + 512 * j, y[j]);
          }
      }
    }
    if (nch > 1) {
      if (!have_w) dma_rows<BC, KC, false>(rsW, g.K, col0, KC, Wreg + W_SLOT);
      dma_rows<BRT, KC, true>(rsA, g.lda, row0, KC, Areg + A_SLOT);
    }
    f32x4 acc[NCT];
#pragma unroll
    for (int j = 0; j < NCT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nch; ++c) {
      // chunk c has landed when only chunk c + 1's loads of THIS wave are outstanding (in-order return): its A and W pieces, or the A
      // pieces alone when chunk 1 of the weights was requested before the barrier (it is then OLDER than chunk 0 of A)
      if (c + 1 < nch) wait_vmcnt((c == 0 && have_w) ? nA : nA + nW);
      else             asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (wave < MW) mma_chunk<BRT, NCT, KC>(Areg + (c & 1) * A_SLOT, Wreg + (c & 1) * W_SLOT, acc);
      if (c + 2 < nch) {
        __syncthreads();
        dma_rows<BC, KC, false>(rsW, g.K, col0, (c + 2) * KC, Wreg + (c & 1) * W_SLOT);
        dma_rows<BRT, KC, true>(rsA, g.lda, row0, (c + 2) * KC, Areg + (c & 1) * A_SLOT);
      }
    }
    // epilogue: lane (lr, lg) holds row 16 rt + lr, columns 16 (ct0 + j) + 4 lg .. + 3
    const int lr = lane & 15, lg = lane >> 4, rt = wave % NRT, ct0 = (wave / NRT) * NCT;
    const int m = row0 + 16 * rt + lr;
    if (wave < MW && m < R) {
#pragma unroll
      for (int j = 0; j < NCT; ++j) {
        const int n = col0 + 16 * (ct0 + j) + 4 * lg;
        f32x4 v = acc[j] + Vec4<T>::load(g.bias + n);
        if (EMODE == 1) {
          v += g.resid_sc1 ? ld_sc1_bf16x4(g.resid + (int64_t)m * D + n) : Vec4<T>::load(g.resid + (int64_t)m * D + n);
          Vec4<float>::store_wt(g.pre_out + (int64_t)m * D + n, v);
        } else {
          if (EMODE == 2) v = gelu_erf4(v);
          Vec4<T>::store_wt(g.out + (int64_t)m * g.ldo + n, v);
        }
      }
    }
    first = false;
  }
}

// One wave per (hypothesis, head) pair, SENTENCE-affine: XCD x (workgroups x, x + 8, ...) takes the pairs of sentences
// [x spx, (x + 1) spx), so the `rep` hypotheses of a sentence -- which read the same encoder K|V rows and, through the slot table, mostly
// the same self-attention cache rows -- meet in one L2.  `rep`: hypotheses per sentence (the cross-attention's a.rep is the same number).
IMT_DEVICE void attn_phase(const imt_attn_decode_args& a, int rep) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = (int)blockIdx.x & 7, nx = (int)gridDim.x >> 3, slot0 = wave * nx + ((int)blockIdx.x >> 3), nslots = FWAVES * nx;
  const int B = a.R / rep, spx = (B + 7) >> 3;
  const int s_lo = xcd * spx, nsent = max(0, min(B, s_lo + spx) - s_lo), per_sent = a.H * rep;
  for (int s = slot0; s < nsent * per_sent; s += nslots) {
    const int sent = s_lo + s / per_sent, h = (s % per_sent) / rep, beam = s % rep;
    attn_decode_wave<T, 64, true, FUSED_UNR>(a, (sent * rep + beam) * a.H + h);
  }
}

template <int D>
__global__ __launch_bounds__(FTHREADS) void decode_fused_kernel(ImtFusedArgs p) {
  typedef FCfg<D> C;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int give_up;
  if (threadIdx.x == 0) give_up = 0;
  __syncthreads();
  GridBar bar{p.bar, 0, &give_up, p.trace};
  if (p.trace && threadIdx.x == 0) p.trace[(int64_t)blockIdx.x * 128 + 127] = __builtin_amdgcn_s_memrealtime();
  const int R = p.R;
  constexpr int d = D;
  const int64_t row3 = (int64_t)p.t_max * 3 * d;
  const float scale = 1.0f / sqrtf((float)p.dh);
  bool w_ready = false;
#pragma unroll 1
  for (int l = 0; l < p.n_layers; ++l) {
    const ImtFusedLayer& L = p.L[l];
    GP g;
    // ---- P1: q|k|v of the new position, straight into the cache row of each hypothesis
    g = GP{};
    g.W = L.wqkv; g.bias = L.bqkv; g.N = 3 * d; g.K = d; g.out = L.cache + (int64_t)p.pos * 3 * d; g.ldo = row3;
    if (l == 0) {
      g.ids = p.ids; g.pos_ids = p.pos_ids; g.type_ids = p.type_ids; g.emb_word = p.emb_word; g.emb_pos = p.emb_pos; g.emb_type = p.emb_type;
      g.vocab = p.vocab; g.max_pos = p.max_pos; g.n_types = p.n_types; g.gamma = p.emb_g; g.beta = p.emb_b; g.norm_out = L.xin;
      gemm_phase<D, 32, 2, 2, 0, D>(g, R, p.eps, smem, w_ready);
    } else {
      g.pre = p.L[l - 1].pre3; g.gamma = p.L[l - 1].g3; g.beta = p.L[l - 1].b3; g.norm_out = L.xin;
      gemm_phase<D, 32, 2, 1, 0, D>(g, R, p.eps, smem, w_ready);
    }
    const T* x_in = L.xin;   // layer input: the embedding LayerNorm (l == 0) or the previous layer's output LayerNorm, stored by P1's column tile 0
    GP g3{};
    g3.W = L.wo; g3.bias = L.bo; g3.N = d; g3.K = d; g3.A = L.ctx1; g3.lda = d; g3.pre_out = L.pre1; g3.resid = x_in; g3.resid_sc1 = true;
    __syncthreads();
    prefetch_w<32, 1, D>(g3, R, smem);   // the weights of P3 travel under the barrier and the self attention
    if (!bar.sync()) return;
    // ---- P2: self attention of the new position over the cache (slot table: beam reordering)
    {
      imt_attn_decode_args a;
      a.dtype = IMT_BF16; a.R = R; a.H = p.H; a.head_dim = p.dh; a.n_keys = p.pos + 1; a.rep = 1;
      a.Q = L.cache + (int64_t)p.pos * 3 * d; a.ldq = row3;
      a.K = L.cache + d; a.V = L.cache + 2 * d; a.ld_row = row3; a.ld_pos = 3 * d;
      a.slots = p.slots; a.ld_slots = p.t_max; a.key_mask = nullptr; a.ld_mask = 0;
      a.O = L.ctx1; a.ldo = d; a.scale = scale; a.reserved = 0;
      attn_phase(a, p.rep);
    }
    if (!bar.sync()) return;
    // ---- P3: output projection + residual -> pre1 (fp32)
    gemm_phase<D, 32, 1, 0, 1, D>(g3, R, p.eps, smem, true);
    GP g4{};
    g4.W = L.wq; g4.bias = L.bq; g4.N = d; g4.K = d; g4.pre = L.pre1; g4.gamma = L.g1; g4.beta = L.b1; g4.norm_out = L.a; g4.out = L.q; g4.ldo = d;
    __syncthreads();
    prefetch_w<32, 1, D>(g4, R, smem);
    if (!bar.sync()) return;
    // ---- P4: LayerNorm -> cross-attention query
    gemm_phase<D, 32, 1, 1, 0, D>(g4, R, p.eps, smem, true);
    GP g6{};
    g6.W = L.wo2; g6.bias = L.bo2; g6.N = d; g6.K = d; g6.A = L.ctx2; g6.lda = d; g6.pre_out = L.pre2; g6.resid = L.a; g6.resid_sc1 = true;
    __syncthreads();
    prefetch_w<32, 1, D>(g6, R, smem);
    if (!bar.sync()) return;
    // ---- P5: cross attention against the sentence's encoder K|V
    {
      imt_attn_decode_args a;
      a.dtype = IMT_BF16; a.R = R; a.H = p.H; a.head_dim = p.dh; a.n_keys = p.Tk; a.rep = p.rep;
      a.Q = L.q; a.ldq = d;
      a.K = L.cross_kv; a.V = L.cross_kv + d; a.ld_row = (int64_t)p.Tk * 2 * d; a.ld_pos = 2 * d;
      a.slots = nullptr; a.ld_slots = 0; a.key_mask = p.enc_mask; a.ld_mask = p.Tk;
      a.O = L.ctx2; a.ldo = d; a.scale = scale; a.reserved = 0;
      // (a form that reads a sentence's K|V once for its `rep` hypotheses -- key segments on 8 waves, partials merged through LDS -- was
      // built and measured: 11.5 against 9.7 us per phase; the phase is two dependent round trips plus shuffles, not L2 traffic)
      attn_phase(a, p.rep);
    }
    if (!bar.sync()) return;
    // ---- P6: output projection + residual -> pre2
    gemm_phase<D, 32, 1, 0, 1, D>(g6, R, p.eps, smem, true);
    GP g7{};
    g7.W = L.w1; g7.bias = L.bf1; g7.N = p.ff; g7.K = d; g7.pre = L.pre2; g7.gamma = L.g2; g7.beta = L.b2; g7.norm_out = L.b; g7.out = L.h; g7.ldo = p.ff;
    __syncthreads();
    prefetch_w<C::BRT_UP, 2, D>(g7, R, smem);
    if (!bar.sync()) return;
    // ---- P7: LayerNorm -> FFN-up + GELU
    gemm_phase<D, C::BRT_UP, 2, 1, 2, D>(g7, R, p.eps, smem, true);
    GP g8{};
    g8.W = L.w2; g8.bias = L.bf2; g8.N = d; g8.K = p.ff; g8.A = L.h; g8.lda = p.ff; g8.pre_out = L.pre3; g8.resid = L.b; g8.resid_sc1 = true;
    __syncthreads();
    prefetch_w<32, 1, C::KC2>(g8, R, smem);
    if (!bar.sync()) return;
    // ---- P8: FFN-down + residual -> pre3 (its LayerNorm: the next layer's P1, or the tail below)
    gemm_phase<D, 32, 1, 0, 1, C::KC2>(g8, R, p.eps, smem, true);
    w_ready = false;
    if (l + 1 < p.n_layers) {
      GP gn{};
      gn.W = p.L[l + 1].wqkv; gn.N = 3 * d; gn.K = d;
      __syncthreads();
      prefetch_w<32, 2, D>(gn, R, smem);
      w_ready = true;
    }
    if (!bar.sync()) return;
  }
  // ---- tail: LayerNorm of the last layer's output -> out
  {
    const ImtFusedLayer& L = p.L[p.n_layers - 1];
    const __amdgpu_buffer_rsrc_t rs = rsrc_of(L.pre3, (int64_t)R * D * 4);
    float gm[C::NJ][8], be[C::NJ][8];
    load_gamma_beta<D>(L.g3, L.b3, gm, be);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int row = (int)blockIdx.x * FWAVES + wave; row < R; row += (int)gridDim.x * FWAVES) {
      bf16x8 y[C::NJ];
      ln_finish<D>(ln_request<D>(rs, row), gm, be, p.eps, y);
#pragma unroll
      for (int j = 0; j < C::NJ; ++j)
        if (group_ok<D>(j)) *reinterpret_cast<bf16x8*>(p.out + (int64_t)row * D + 8 * lane + 512 * j) = y[j];
    }
  }
}

}  // namespace

int64_t imt_decode_fused_layer_bytes(int r_max, int d, int ff) {
  const int64_t R = r_max;
  auto up = [](int64_t b) { return (b + 255) & ~(int64_t)255; };   // every buffer starts on a 256-byte boundary
  return 6 * up(R * d * 2) + up(R * ff * 2) + 3 * up(R * d * 4);
}

bool imt_decode_fused_shape(int d, int heads, int ff, int n_layers) {
  if (n_layers > IMT_FUSED_MAX_LAYERS || heads * 64 != d) return false;
  if (d == 512) return ff >= 512 && ff % 512 == 0;
  if (d == 768) return ff >= 768 && ff % 384 == 0 && ff % 64 == 0;
  return false;
}

bool imt_decode_fused_enabled() {
  const char* e = getenv("IMT_DECODE_FUSED");   // read per call (a step costs one getenv): tests flip it inside one process
  return e ? atoi(e) != 0 : true;
}

int imt_decode_fused_launch(const ImtFusedArgs& a, hipStream_t st) {
  static int grid = 0;
  if (grid == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
      imt_set_error("decode_fused: cannot read the CU count");
      return IMT_ERR_LAUNCH;
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(decode_fused_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, FCfg<512>::LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(decode_fused_kernel<768>), hipFuncAttributeMaxDynamicSharedMemorySize, FCfg<768>::LDS) != hipSuccess) {
      imt_set_error("decode_fused: cannot raise the LDS limit");
      return IMT_ERR_LAUNCH;
    }
    grid = (cus < 256 ? cus : 256) & ~7;   // one workgroup per CU (128 / 144 KiB of LDS each): all resident, a multiple of the 8 XCDs
    if (grid < 8) { imt_set_error("decode_fused: %d CUs", cus); grid = 0; return IMT_ERR_LAUNCH; }
  }
  // barrier counters start at zero in every launch; the status word (last) is sticky
  if (hipMemsetAsync(a.bar, 0, (IMT_FUSED_BAR_WORDS - 1) * sizeof(unsigned), st) != hipSuccess) {
    imt_set_error("decode_fused: memset failed");
    return IMT_ERR_LAUNCH;
  }
  static const bool want_trace = getenv("IMT_DECODE_TRACE") != nullptr;
  static unsigned long long* tbuf = nullptr;
  ImtFusedArgs b = a;
  if (want_trace) {
    if (!tbuf && hipMalloc(&tbuf, 256 * 128 * 8) != hipSuccess) tbuf = nullptr;
    if (tbuf) (void)hipMemsetAsync(tbuf, 0, 256 * 128 * 8, st);
    b.trace = tbuf;
  }
  {
    ImtProfScope prof("decode_fused", 0, 0, st);
    if (a.d == 768) hipLaunchKernelGGL(decode_fused_kernel<768>, dim3(grid), dim3(FTHREADS), FCfg<768>::LDS, st, b);
    else            hipLaunchKernelGGL(decode_fused_kernel<512>, dim3(grid), dim3(FTHREADS), FCfg<512>::LDS, st, b);
  }
  IMT_CHECK_LAUNCH();
  if (want_trace && tbuf) {   // tuning aid: synchronises and prints the mean length of every phase and of every barrier
    static unsigned long long h[256 * 128];
    static int calls = 0;
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, tbuf, sizeof(h), hipMemcpyDeviceToHost);
    if (++calls % 40 == 0) {
      const int nb = 8 * a.n_layers;
      fprintf(stderr, "[decode_fused pos %d rows %d] phase (work until arrival) / barrier (arrival -> departure), mean over %d workgroups, us:\n", a.pos, a.R, grid);
      for (int r = 0; r < nb; ++r) {
        double work = 0, wait = 0;
        for (int g = 0; g < grid; ++g) {
          const unsigned long long* t = h + (int64_t)g * 128;
          const unsigned long long start = r == 0 ? t[127] : t[2 * r - 1];
          work += (double)(t[2 * r] - start); wait += (double)(t[2 * r + 1] - t[2 * r]);
        }
        fprintf(stderr, "  L%d P%d  %6.2f / %5.2f\n", r / 8, r % 8 + 1, work / grid / 100.0, wait / grid / 100.0);
      }
    }
  }
  return IMT_OK;
}
