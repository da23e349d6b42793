// MFMA GEMM for every nn.Linear forward/backward on the path (see include/imt_hip.h, imt_gemm).
//
// Tile: 128 x 128 output per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA 16x16 tiles),
// K advanced 128 BYTES per tile row (64 bf16 / 32 fp32) so the LDS geometry is the same for both types.
// Operands are staged global -> registers -> LDS (16-B chunks, XOR-swizzled, double-buffered: the loads of
// tile t+1 are issued before the MFMAs of tile t and written to the other buffer after them; one barrier
// per K tile).  K-contiguous operands are read back with ds_read_b128; K-strided operands (the N-contiguous
// B of NN, both operands of TN) with the gfx950 transposed LDS read ds_read_b64_tr_b16 (bf16) or scalar
// reads (fp32) -- no operand is ever transposed through HBM.
// The product is accumulated as C^T tiles (mfma(Bfrag, Afrag)) so each lane owns 4 consecutive n of one m:
// bias / residual / aux accesses and the C store are 8- or 16-byte vectors.
#include "mma.hpp"

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;

template <typename T, bool KCONTIG> struct Stage {
  // tile bytes = 16 KiB either way
  static constexpr int EPC = 16 / sizeof(T);                       // elements per 16-B chunk
  static constexpr int BK = 128 / sizeof(T);                       // K elements per tile
  static constexpr int RB = KCONTIG ? 128 : 128 * sizeof(T);       // LDS row bytes
  static constexpr int CPR = RB / 16;                              // chunks per row
  u32x4 r[4];

  // KCONTIG: operand[rows = M or N][K], tile rows = 128 operand rows, row chunk c covers k0 + c*EPC
  // else   : operand[K][cols = M or N], tile rows = BK k-rows, row chunk c covers col0 + c*EPC
  IMT_DEVICE void load(const T* __restrict__ base, int64_t ld, int row0, int nrows, int k0, int K) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i * NTHREADS + tid;
      const int tr = q / CPR, c = q % CPR;
      int64_t grow, gcol;
      bool ok;
      if (KCONTIG) { grow = row0 + tr; gcol = k0 + c * EPC; ok = (grow < nrows) && (gcol < K); }
      else         { grow = k0 + tr;   gcol = row0 + c * EPC; ok = (grow < K) && (gcol < nrows); }
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
      const int tr = q / CPR, c = q % CPR;
      *reinterpret_cast<u32x4*>(tile + tile_off<RB>(tr, c)) = r[i];
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
};

template <typename T, int LAYOUT>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                                        int64_t ldb, int M, int N, int K, int k_per_split,
                                                        EpiParams ep) {
  constexpr bool A_KC = (LAYOUT != IMT_TN);  // A is K-contiguous for NT, NN
  constexpr bool B_KC = (LAYOUT == IMT_NT);  // B is K-contiguous for NT only
  typedef Stage<T, A_KC> SA;
  typedef Stage<T, B_KC> SB;
  typedef typename Frag<T>::type frag_t;
  constexpr int BK = SA::BK;
  constexpr int KSTEP = Frag<T>::KSTEP;  // k per fragment step (64 bytes)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // buffer b: A tile at smem + b*32768, B tile 16 KiB after it

  // XCD-aware block order (T1): consecutive ids on one XCD walk neighbouring tiles.
  const int nbx = (N + BN - 1) / BN, nby = (M + BM - 1) / BM;
  const int nwg = nbx * nby;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (bid / nbx) * BM, n0 = (bid % nbx) * BN;
  const int kbeg = blockIdx.y * k_per_split;
  const int kend = min(K, kbeg + k_per_split);

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int lr = lane & 15, lg = lane >> 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  SA sa; SB sb;
  const int nt = (kend - kbeg + BK - 1) / BK;
  if (nt > 0) {
    sa.load(A, lda, m0, M, kbeg, kend);
    sb.load(B, ldb, n0, N, kbeg, kend);
    sa.store(smem);
    sb.store(smem + 16384);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) {
      sa.load(A, lda, m0, M, kbeg + (t + 1) * BK, kend);
      sb.load(B, ldb, n0, N, kbeg + (t + 1) * BK, kend);
    }
    const char* ta = smem + cur * 32768;
    const char* tb = ta + 16384;
#pragma unroll
    for (int s = 0; s < BK / KSTEP; ++s) {
      frag_t fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (A_KC) fa[i] = lds_frag_kcontig<T, SA::RB>(ta, wm + 16 * i, 4 * s);
        else      fa[i] = KStrided<T, SA::RB>::load(ta, s * KSTEP, wm + 16 * i);
        if (B_KC) fb[i] = lds_frag_kcontig<T, SB::RB>(tb, wn + 16 * i, 4 * s);
        else      fb[i] = KStrided<T, SB::RB>::load(tb, s * KSTEP, wn + 16 * i);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma16(acc[i][j], fb[j], fa[i]);  // C^T tile: rows <- n, cols <- m
    }
    if (t + 1 < nt) {
      sa.store(smem + (cur ^ 1) * 32768);
      sb.store(smem + (cur ^ 1) * 32768 + 16384);
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  const float alpha = ep.alpha_dev ? ep.alpha * ep.alpha_dev[0] : ep.alpha;
  // lane owns C[m = m0+wm+16i+lr][n = n0+wn+16j+4lg .. +3]
  const T* bias = reinterpret_cast<const T*>(ep.bias);
  const T* resid = reinterpret_cast<const T*>(ep.resid);
  T* aux = reinterpret_cast<T*>(ep.aux);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm + 16 * i + lr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn + 16 * j + 4 * lg;
      if (n >= N) continue;
      f32x4 v = acc[i][j] * alpha;
      const bool full = (n + 3 < N);
      if (ep.atomic) {
        float* c = reinterpret_cast<float*>(ep.C) + (int64_t)m * ep.ldc + n;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < N) atomicAdd(c + e, v[e]);
        continue;
      }
      if (full) {
        if (bias) v += Vec4<T>::load(bias + n);
        if (ep.aux_mode == IMT_AUX_GELU_FWD) {
          Vec4<T>::store(aux + (int64_t)m * ep.ldaux + n, v);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        } else if (ep.aux_mode == IMT_AUX_DGELU) {
          f32x4 z = Vec4<T>::load(aux + (int64_t)m * ep.ldaux + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(z[e]);
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
          Vec4<float>::store(c, v);
        } else {
          T* c = reinterpret_cast<T*>(ep.C) + (int64_t)m * ep.ldc + n;
          if (ep.accumulate) v += Vec4<T>::load(c);
          Vec4<T>::store(c, v);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= N) break;
          float x = v[e];
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
}

template <typename T, int LAYOUT>
int launch(const imt_gemm_args* a, const EpiParams& ep, int splits, int k_per_split, hipStream_t st) {
  const int nbx = imt_cdiv(a->N, BN), nby = imt_cdiv(a->M, BM);
  dim3 grid(nbx * nby, splits);
  static bool attr_set = false;
  auto kern = gemm_kernel<T, LAYOUT>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr_set = true;
  }
  static const char* const kinds[2][3] = {{"gemm_f32_nt", "gemm_f32_nn", "gemm_f32_tn"}, {"gemm_bf16_nt", "gemm_bf16_nn", "gemm_bf16_tn"}};
  const double es = sizeof(T), esc = ep.c_f32 ? 4.0 : es;
  ImtProfScope prof(kinds[sizeof(T) == 2][LAYOUT], 2.0 * a->M * a->N * a->K,
                    ((double)a->M * a->K + (double)a->N * a->K) * es + (double)a->M * a->N * esc, st);
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), 65536, st, reinterpret_cast<const T*>(a->A), a->lda,
                     reinterpret_cast<const T*>(a->B), a->ldb, a->M, a->N, a->K, k_per_split, ep);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

template <typename T> int dispatch(const imt_gemm_args* a, const EpiParams& ep, int splits, int kps, hipStream_t st) {
  switch (a->layout) {
    case IMT_NT: return launch<T, IMT_NT>(a, ep, splits, kps, st);
    case IMT_NN: return launch<T, IMT_NN>(a, ep, splits, kps, st);
    case IMT_TN: return launch<T, IMT_TN>(a, ep, splits, kps, st);
  }
  imt_set_error("imt_gemm: bad layout %d", a->layout);
  return IMT_ERR_BAD_ARG;
}

}  // namespace

extern "C" int imt_gemm(const imt_gemm_args* a, void* stream) {
  IMT_CHECK_ARG(a != nullptr, "imt_gemm: null args");
  IMT_CHECK_ARG(a->dtype == IMT_F32 || a->dtype == IMT_BF16, "imt_gemm: bad dtype %d", a->dtype);
  IMT_CHECK_ARG(a->M >= 0 && a->N >= 0 && a->K >= 0, "imt_gemm: negative dims");
  if (a->M == 0 || a->N == 0) return IMT_OK;
  IMT_CHECK_ARG(a->A && a->B && a->C, "imt_gemm: null operand");
  const int al = (a->dtype == IMT_BF16) ? 8 : 4;
  IMT_CHECK_ARG(a->lda % al == 0 && a->ldb % al == 0, "imt_gemm: lda/ldb must be multiples of %d (16-B rows)", al);
  IMT_CHECK_ARG(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "imt_gemm: A/B must be 16-B aligned");
  // contiguous extents of the vector loads must be chunk multiples
  const int a_inner = (a->layout == IMT_TN) ? a->M : a->K;
  const int b_inner = (a->layout == IMT_NT) ? a->K : a->N;
  IMT_CHECK_ARG(a_inner % al == 0 && b_inner % al == 0, "imt_gemm: inner extents (%d,%d) must be multiples of %d",
                a_inner, b_inner, al);
  IMT_CHECK_ARG(a->ldc % 4 == 0, "imt_gemm: ldc must be a multiple of 4");
  const int c_f32 = (a->c_dtype == IMT_F32);
  IMT_CHECK_ARG(c_f32 || a->c_dtype == a->dtype, "imt_gemm: c_dtype must be f32 or dtype");
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
  EpiParams ep;
  ep.C = a->C; ep.ldc = a->ldc; ep.c_f32 = c_f32; ep.accumulate = a->accumulate;
  ep.bias = a->bias; ep.resid = a->resid; ep.ldr = a->ldr;
  ep.aux = a->aux; ep.ldaux = a->ldaux; ep.aux_mode = a->aux_mode;
  ep.atomic = (splits > 1);
  ep.alpha = a->alpha; ep.alpha_dev = a->alpha_dev;
  ep.drop_thresh = dropout_thresh(a->dropout_p);
  ep.inv_keep = a->dropout_p > 0.f ? 1.0f / (1.0f - a->dropout_p) : 1.0f;
  ep.seed = a->dropout_seed;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a->dtype == IMT_F32) return dispatch<float>(a, ep, splits, kps, st);
  return dispatch<bf16_t>(a, ep, splits, kps, st);
}
