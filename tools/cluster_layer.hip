// MOCK of DESIGN.md section 7's batch-element-persistent encoder stack (a performance experiment, not part of the library):
// one launch runs L post-LN encoder layers (d = 512, 8 heads of 64, ff = 2048, T = 128 tokens per sentence, no masks, no dropout).
// A CLUSTER of 4 workgroups (one per CU, all on ONE XCD: block ids congruent mod 8) owns a sentence through the whole stack;
// member m computes a quarter of the columns of every product:
//   P1  q|k|v of heads 2m, 2m+1 (128 x 384 slice, K = 512)  -> attention of those two heads -> context slice      [hand-off 1: 32 KB]
//   P3  output projection slice (128 x 128, K = 512) + bias + residual -> row sums                                [hand-off 2:  1 KB]
//       LayerNorm with the four members' sums -> y1 slice                                                          [hand-off 3: 32 KB]
//   P4  FFN-up slice (128 x 512 in two passes of 256, K = 512) + bias -> pre-activation (saved) -> GELU -> h slice [hand-off 4: 128 KB]
//   P5  FFN-down slice (128 x 128, K = 2048) + bias + residual -> row sums                                        [hand-off 5:  1 KB]
//       LayerNorm -> y2 slice = the next layer's input                                                             [hand-off 6: 32 KB]
// Every tensor the library's backward would read is stored (q|k|v, context, y1, pre-activation, h, y2).  Hand-offs follow the
// micro-architecture guide's Guideline 16 and tools/probe_handoff.hip: sc1 stores, every wave drains, one lane stores the
// flag; one wave polls the three peers (relaxed, bounded, s_sleep), barrier, every load of handed-off bytes is sc1 (LDS-DMA
// with aux = sc1 for operand tiles).  Weight tiles of the next product are requested BEFORE the wait (they depend on nobody).
// Driven and checked against torch by tools/cluster_layer.py.
#include "../imagetranslate_amd/csrc/mma.hpp"

typedef bf16_t T;
#ifndef XCD_LOCAL
#define XCD_LOCAL 0   // 1: hand-offs through the XCD's own L2 (plain stores, plain operand loads): valid only while a cluster sits on one XCD
#endif
typedef Frag<T>::type frag_t;
typedef __attribute__((address_space(1))) unsigned gu32;
constexpr int RB = 128, TT = 128, D = 512, FF = 2048, MAXL = 8;
constexpr int MISC = 147456;   // LDS: the operand ring (2 x 64, 3 x 48 or 4 x 32 KiB), then 4 KiB of row partials + 1 KiB of row statistics
constexpr int LDS_BYTES = MISC + 4096 + 1024;

struct LayerW { const T *wqkv, *bqkv, *wo, *bo, *g1, *b1, *w1, *bf1, *w2, *bf2, *g2, *b2; };
struct LayerBuf { T *qkv, *ctx, *y1, *pre, *h, *y2; float *st1, *st2; };   // st*: [4 members][rows][2] floats
struct ClP { unsigned long long* trace; const T* x0; LayerW w[MAXL]; LayerBuf buf[MAXL]; unsigned* flags; unsigned* status; int layers; int rows; unsigned epoch0; float scale; };

#if XCD_LOCAL
#define STORE_HANDOFF(p, v) Vec4<T>::store(p, v)
#else
#define STORE_HANDOFF(p, v) Vec4<T>::store_wt(p, v)
#endif
IMT_DEVICE unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
IMT_DEVICE void st_rlx(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
IMT_DEVICE f32x2 ld_sc1_f2(const float* p) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(f32x2, v);
}
IMT_DEVICE void st_sc1_f2(float* p, f32x2 v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the K loop of one 128 x (64 NJ) slice: A rows = the sentence's 128 tokens, W rows = the slice's output columns (K-contiguous)
template <int NJ, int S> struct Slice {   // S ring stages of (16 + 8 NJ) KiB
  static constexpr int STAGE = 16384 + NJ * 8192, LPT = 2 + NJ;
  __amdgpu_buffer_rsrc_t ra, rw;
  int va[2], vw[NJ];
  int wave, lane;
  IMT_DEVICE void init_w(const T* W, int64_t ldw, int64_t wrows, int wrow0, int wstride) {
    wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); lane = threadIdx.x & 63;
    rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(W), 0, (int)(wrows * ldw * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int pw = wave + 8 * i, ti = pw >> 4, pi = pw & 15;
      const int tr = 8 * pi + (lane >> 3), c = (lane & 7) ^ swz<RB>(tr);
      vw[i] = (int)(((int64_t)(wrow0 + ti * wstride + tr) * ldw + c * 8) * 2);
    }
  }
  IMT_DEVICE void init_a(const T* A, int64_t lda) {
    ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(A), 0, (int)(TT * lda * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pi = wave + 8 * i;
      const int tr = 8 * pi + (lane >> 3), c = (lane & 7) ^ swz<RB>(tr);
      va[i] = (int)(((int64_t)tr * lda + c * 8) * 2);
    }
  }
  IMT_DEVICE void issue_w(char* stage, int t) const {
#pragma unroll
    for (int i = 0; i < NJ; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(stage + 16384 + (wave + 8 * i) * 1024), 16,
                                               vw[i] + t * 128, 0, 0, 0);
  }
  IMT_DEVICE void issue_a(char* stage, int t) const {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(stage + (wave + 8 * i) * 1024), 16,
                                               va[i] + t * 128, 0, 0, XCD_LOCAL ? 0 : 16 /* sc1 */);
  }
  // W tile 0 was requested into stage 0 by the caller (before it waited for the peers); acc += A W^T over K
  IMT_DEVICE void run(char* smem, int K, f32x4 (&acc)[4][NJ]) const {
    const int nt = K >> 6;
    const int wm = (wave >> 2) * 64, c0 = (wave & 3) * 16 * NJ;
    issue_a(smem, 0);
#pragma unroll
    for (int u = 1; u <= S - 2; ++u)
      if (u < nt) { issue_a(smem + u * STAGE, u); issue_w(smem + u * STAGE, u); }
    for (int t = 0; t < nt; ++t) {
      // tile t has landed when at most the S - 2 younger tiles' loads are outstanding (in-order return); the tail drains
      if (S > 2 && t + S - 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * LPT) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      const int tn = t + S - 1;   // its stage was read in iteration t - 1: free after the barrier
      if (tn < nt) { issue_a(smem + (tn % S) * STAGE, tn); issue_w(smem + (tn % S) * STAGE, tn); }
      const char* st = smem + (t % S) * STAGE;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        frag_t fa[4], fb[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = c0 + 16 * j;
          fb[j] = lds_frag_kcontig<T, RB>(st + (1 + (c >> 7)) * 16384, c & 127, 4 * s);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = lds_frag_kcontig<T, RB>(st, wm + 16 * i, 4 * s);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) mma16(acc[i][j], fb[j], fa[i]);  // C^T tile: lane (r, gq) holds row wm+16i+r, cols c0+16j+4gq+e
      }
    }
    __syncthreads();
  }
};

template <int NJ> IMT_DEVICE void zero(f32x4 (&acc)[4][NJ]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
}

struct Cluster {
  unsigned *flags, *status; int cluster, member, wg; int* give_up;
  // every wave's stores have to be out before the flag: drain, barrier, one lane publishes
  IMT_DEVICE void publish(unsigned epoch) const {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) st_rlx(flags + wg * 32, epoch);
  }
  // returns false when the launch is being abandoned (a bounded wait ran out somewhere): the caller returns
  IMT_DEVICE bool wait(unsigned epoch) const {
    const int lane = threadIdx.x & 63;
    if ((threadIdx.x >> 6) == 0) {
      bool ok = false;
      for (unsigned spins = 0; spins < 2000000u; ++spins) {
        bool all = true;
        if (lane < 4 && lane != member) all = ld_rlx(flags + (cluster * 4 + lane) * 32) >= epoch;
        if (__all(all)) { ok = true; break; }
        if (ld_rlx(status) != 0) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok && lane == 0) { st_rlx(status, 0x1000u + epoch); *give_up = 1; }
    }
    __syncthreads();
    return *give_up == 0;
  }
};

// residual + bias, row sums through LDS, exchange with the peers, LayerNorm, store the slice (sc1).  v: this wave's 64 x 32 block.
IMT_DEVICE bool resid_ln(f32x4 (&v)[4][2], const T* bias, const T* resid /* [rows][512] */, const T* gamma, const T* beta, float* st_buf,
                         T* out, int64_t row0, int rows, const Cluster& cl, unsigned epoch, char* smem) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, gq = lane >> 4;
  const int wm = (wave >> 2) * 64, c0 = (wave & 3) * 32, col0 = cl.member * 128;
  float* part = reinterpret_cast<float*>(smem + MISC);          // [128 rows][4 column groups][2]
  float* rstat = reinterpret_cast<float*>(smem + MISC + 4096);  // [128 rows][2] = mean, rstd
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = wm + 16 * i + r;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = col0 + c0 + 16 * j + 4 * gq;
      const f32x4 x = v[i][j] + Vec4<T>::load(bias + c) + Vec4<T>::load(resid + (row0 + m) * D + c);
      v[i][j] = x;
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1 += x[e]; s2 += x[e] * x[e]; }
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    if (gq == 0) { part[(m * 4 + (wave & 3)) * 2] = s1; part[(m * 4 + (wave & 3)) * 2 + 1] = s2; }
  }
  __syncthreads();
  float my1 = 0.f, my2 = 0.f;
  if (threadIdx.x < TT) {
    const float* q = part + threadIdx.x * 8;
    my1 = (q[0] + q[2]) + (q[4] + q[6]); my2 = (q[1] + q[3]) + (q[5] + q[7]);
    st_sc1_f2(st_buf + ((int64_t)cl.member * rows + row0 + threadIdx.x) * 2, f32x2{my1, my2});
  }
  cl.publish(epoch);
  if (!cl.wait(epoch)) return false;
  if (threadIdx.x < TT) {
    float t1 = 0.f, t2 = 0.f;  // fixed order over the members: every member computes bit-identical statistics
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
      f32x2 o = f32x2{my1, my2};
      if (mm != cl.member) o = ld_sc1_f2(st_buf + ((int64_t)mm * rows + row0 + threadIdx.x) * 2);
      t1 += o[0]; t2 += o[1];
    }
    const float mean = t1 * (1.f / D), var = fmaxf(t2 * (1.f / D) - mean * mean, 0.f);
    rstat[threadIdx.x * 2] = mean; rstat[threadIdx.x * 2 + 1] = rsqrtf(var + 1e-12f);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = wm + 16 * i + r;
    const float mean = rstat[m * 2], rstd = rstat[m * 2 + 1];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = col0 + c0 + 16 * j + 4 * gq;
      const f32x4 y = (v[i][j] - mean) * rstd * Vec4<T>::load(gamma + c) + Vec4<T>::load(beta + c);
      STORE_HANDOFF(out + (row0 + m) * D + c, y);
    }
  }
  return true;
}

#define STAMP(k) do { if (p.trace && threadIdx.x == 0) p.trace[((int64_t)blockIdx.x * MAXL + l) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__global__ __launch_bounds__(512, 1) void cluster_layers(ClP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int give_up;
  const int bid = blockIdx.x, xcd = bid & 7, q = bid >> 3;
  Cluster cl;
  cl.flags = p.flags; cl.status = p.status; cl.cluster = xcd * (gridDim.x >> 5) + (q >> 2); cl.member = q & 3;
  cl.wg = cl.cluster * 4 + cl.member; cl.give_up = &give_up;
  if (threadIdx.x == 0) give_up = 0;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 15, gq = lane >> 4;
  const int m4 = cl.member;
  const int64_t row0 = (int64_t)cl.cluster * TT;   // the sentence's first token row
  const int wm = (wave >> 2) * 64;
  unsigned epoch = p.epoch0;
  const T* x = p.x0;
  __syncthreads();

#pragma unroll 1
  for (int l = 0; l < p.layers; ++l) {
    const LayerW& w = p.w[l];
    const LayerBuf& o = p.buf[l];
    // ------------------------------------------------------------ P1: q|k|v of two heads, attention of those heads
    {
      STAMP(0);
      Slice<6, 2> g;
      g.init_w(w.wqkv, D, 3 * D, m4 * 128, D);
      g.issue_w(smem, 0);
      if (l > 0 && !cl.wait(epoch)) return;   // the peers' y2 slices of the previous layer
      STAMP(1);
      g.init_a(x + row0 * D, D);
      f32x4 acc[4][6];
      zero<6>(acc);
      g.run(smem, D, acc);
      STAMP(2);
      const int c0 = (wave & 3) * 96;
      char* tiles = smem;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int c = c0 + 16 * j + 4 * gq;
        const int which = c >> 7, hh = (c >> 6) & 1, dim = c & 63;
        const int gcol = which * D + m4 * 128 + (c & 127);
        const f32x4 bv = Vec4<T>::load(w.bqkv + gcol);
        char* tl = tiles + (which * 2 + hh) * 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = wm + 16 * i + r;
          const f32x4 v = acc[i][j] + bv;
          const bf16x4 w4 = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          *reinterpret_cast<bf16x4*>(o.qkv + (row0 + m) * (3 * D) + gcol) = w4;
          *reinterpret_cast<bf16x4*>(tl + tile_off<RB>(m, dim >> 3) + ((dim & 7) << 1)) = w4;
        }
      }
      __syncthreads();
      const int q0 = wave * 16, i = q0 + r;
#pragma unroll 1
      for (int hh = 0; hh < 2; ++hh) {
        const int h = 2 * m4 + hh;
        const char* Qs = tiles + (0 * 2 + hh) * 16384;
        const char* Ks = tiles + (1 * 2 + hh) * 16384;
        const char* Vs = tiles + (2 * 2 + hh) * 16384;
        frag_t qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = lds_frag_kcontig<T, RB>(Qs, q0, 4 * ks);
        f32x4 s[8];
        float tmax = -INFINITY;
#pragma unroll
        for (int n8 = 0; n8 < 8; ++n8) {
          s[n8] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) mma16(s[n8], lds_frag_kcontig<T, RB>(Ks, 16 * n8, 4 * ks), qf[ks]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { s[n8][e] *= p.scale; tmax = fmaxf(tmax, s[n8][e]); }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        float psum = 0.f;
#pragma unroll
        for (int n8 = 0; n8 < 8; ++n8)
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float pv = __expf(s[n8][e] - tmax); psum += pv; s[n8][e] = pv; }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        f32x4 ov[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ov[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const frag_t pf = acc_pair_to_frag(s[2 * u], s[2 * u + 1]);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) mma16(ov[dt], lds_frag_kperm_bf16<RB>(Vs, 32 * u, 16 * dt), pf);
        }
        const float inv_l = 1.0f / psum;
        T* Ob = o.ctx + (row0 + i) * D + h * 64;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) STORE_HANDOFF(Ob + 16 * dt + 4 * gq, ov[dt] * inv_l);
      }
      ++epoch;
      cl.publish(epoch);   // hand-off 1: context slices (the barrier inside also frees the tiles)
      STAMP(3);
    }
    // ------------------------------------------------------------ P3: output projection slice + residual + LayerNorm
    {
      Slice<2, 4> g;
      g.init_w(w.wo, D, D, m4 * 128, 0);
      g.issue_w(smem, 0);
      if (!cl.wait(epoch)) return;
      STAMP(4);
      g.init_a(o.ctx + row0 * D, D);
      f32x4 acc[4][2];
      zero<2>(acc);
      g.run(smem, D, acc);
      STAMP(5);
      ++epoch;   // hand-off 2 inside (row sums), 3 after (y1 slices)
      if (!resid_ln(acc, w.bo, x, w.g1, w.b1, o.st1, o.y1, row0, p.rows, cl, epoch, smem)) return;
      ++epoch;
      cl.publish(epoch);
      STAMP(6);
    }
    // ------------------------------------------------------------ P4: FFN-up slice (two passes of 256 columns), GELU
    {
      Slice<4, 3> g;
      g.init_w(w.w1, D, FF, m4 * 512, 128);
      g.issue_w(smem, 0);
      if (!cl.wait(epoch)) return;
      STAMP(7);
      g.init_a(o.y1 + row0 * D, D);
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        if (pass) { g.init_w(w.w1, D, FF, m4 * 512 + 256, 128); g.issue_w(smem, 0); }
        f32x4 acc[4][4];
        zero<4>(acc);
        g.run(smem, D, acc);
        const int c0 = (wave & 3) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = m4 * 512 + pass * 256 + c0 + 16 * j + 4 * gq;
          const f32x4 bv = Vec4<T>::load(w.bf1 + c);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = wm + 16 * i + r;
            const f32x4 z = acc[i][j] + bv;
            Vec4<T>::store(o.pre + (row0 + m) * FF + c, z);
            STORE_HANDOFF(o.h + (row0 + m) * FF + c, gelu_erf4(z));
          }
        }
      }
      ++epoch;
      cl.publish(epoch);   // hand-off 4: h slices
      STAMP(8);
    }
    // ------------------------------------------------------------ P5: FFN-down slice + residual + LayerNorm
    {
      Slice<2, 4> g;
      g.init_w(w.w2, FF, D, m4 * 128, 0);
      g.issue_w(smem, 0);
      if (!cl.wait(epoch)) return;
      STAMP(9);
      g.init_a(o.h + row0 * FF, FF);
      f32x4 acc[4][2];
      zero<2>(acc);
      g.run(smem, FF, acc);
      STAMP(10);
      ++epoch;   // hand-off 5 inside, 6 after
      if (!resid_ln(acc, w.bf2, o.y1, w.g2, w.b2, o.st2, o.y2, row0, p.rows, cl, epoch, smem)) return;
      ++epoch;
      cl.publish(epoch);
      STAMP(11);
    }
    x = o.y2;
  }
}

// ptrs: x0, then per layer 12 weight pointers (LayerW order) and 8 buffer pointers (LayerBuf order)
// trace: nullptr or [256 workgroups][MAXL][16] 100-MHz stamps (0 layer start, 1 input arrived, 2 q|k|v product, 3 attention + publish,
// 4 contexts arrived, 5 projection, 6 LayerNorm incl. its exchange + publish, 7 y1 arrived, 8 FFN-up + publish, 9 h arrived, 10 FFN-down, 11 LayerNorm + publish)
extern "C" int cl_run(const void* const* ptrs, int layers, int sentences, unsigned* flags, unsigned* status, unsigned epoch0, void* stream,
                      unsigned long long* trace) {
  if (layers < 1 || layers > MAXL || sentences != 64) return 1;   // 64 clusters x 4 = 256 workgroups, one per CU: all resident
  ClP p;
  p.trace = trace;
  p.x0 = (const T*)ptrs[0];
  for (int l = 0; l < layers; ++l) {
    const void* const* q = ptrs + 1 + 20 * l;
    p.w[l] = LayerW{(const T*)q[0], (const T*)q[1], (const T*)q[2], (const T*)q[3], (const T*)q[4], (const T*)q[5],
                    (const T*)q[6], (const T*)q[7], (const T*)q[8], (const T*)q[9], (const T*)q[10], (const T*)q[11]};
    p.buf[l] = LayerBuf{(T*)q[12], (T*)q[13], (T*)q[14], (T*)q[15], (T*)q[16], (T*)q[17], (float*)q[18], (float*)q[19]};
  }
  p.flags = flags; p.status = status; p.layers = layers; p.rows = sentences * TT; p.epoch0 = epoch0; p.scale = 0.125f;
  static bool once = false;
  if (!once) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_layers), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return 2;
    once = true;
  }
  hipLaunchKernelGGL(cluster_layers, dim3(sentences * 4), dim3(512), LDS_BYTES, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
extern "C" int cl_epochs_per_launch(int layers) { return 6 * layers; }
