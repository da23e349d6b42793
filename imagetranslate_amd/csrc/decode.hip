// Incremental-decoding kernels: single-query attention over a slot-addressed K/V cache, and the on-device beam
// step (log-softmax + length-penalised top-k + the reference's bookkeeping, src/seq_gen.py:193-227).
// Both are HBM/latency-bound byte and index work: no MFMA here, only coalesced 16-byte row reads.
#include "common.hpp"
#include "decode_attn.hpp"

namespace {

// ------------------------------------------------------------------------------------------------ beam step
// (score desc, index asc) ordering: a "better" than b
IMT_DEVICE bool better(float sa, int ia, float sb, int ib) { return sa > sb || (sa == sb && ia < ib); }

IMT_DEVICE void wave_best(float& s, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float so = __shfl_xor(s, o, 64);
    const int io = __shfl_xor(i, o, 64);
    if (better(so, io, s, i)) { s = so; i = io; }
  }
}

constexpr int BEAM_THREADS = 256;
constexpr int MAX_BEAM = 32;

// Stage 1: one workgroup per input hypothesis row: log-sum-exp, then the row's `beam` best continuations.
__global__ __launch_bounds__(BEAM_THREADS) void beam_row_topk_kernel(imt_beam_args a) {
  __shared__ float red_f[BEAM_THREADS / 64];
  __shared__ float red_g[BEAM_THREADS / 64];
  __shared__ int red_i[BEAM_THREADS / 64];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = r / a.rep;
  const float* x = a.logits + (int64_t)r * a.ld;
  const float cur = a.scores_in[r];
  const bool use_pen = a.beam > 1;
  const float pen = use_pen ? powf((a.sizes_in[r] + 6.0f) / 6.0f, a.len_penalty_ratio) : 1.0f;
  const bool masked = a.eos_in[r] || (a.step > 1 && a.max_lens[b] < (int64_t)a.step + 1);
  float* cs = a.cand_scores + (int64_t)r * a.beam;
  int* ci = a.cand_idx + (int64_t)r * a.beam;
  if (masked) {  // all V continuations score the same: lowest indices win
    const float s = use_pen ? (cur + 0.0f) / pen : cur + 0.0f;
    for (int t = tid; t < a.beam; t += BEAM_THREADS) { cs[t] = s; ci[t] = t; }
    return;
  }
  // log-sum-exp
  float mx = -INFINITY;
  for (int v = tid; v < a.V; v += BEAM_THREADS) mx = fmaxf(mx, x[v]);
  mx = wave_max(mx);
  if (lane == 0) red_f[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
  float sum = 0.f;
  for (int v = tid; v < a.V; v += BEAM_THREADS) sum += __expf(x[v] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red_g[wv] = sum;
  __syncthreads();
  const float lse = mx + logf(red_g[0] + red_g[1] + red_g[2] + red_g[3]);
  // `beam` selection rounds; round t admits only elements ordered after the previous pick
  float ps = INFINITY; int pi = -1;
  for (int t = 0; t < a.beam; ++t) {
    float bs = -INFINITY; int bi = 0x7fffffff;
    for (int v = tid; v < a.V; v += BEAM_THREADS) {
      const float lp = x[v] - lse;
      const float s = use_pen ? (cur + lp) / pen : cur + lp;
      const bool eligible = (t == 0) || s < ps || (s == ps && v > pi);
      if (eligible && better(s, v, bs, bi)) { bs = s; bi = v; }
    }
    wave_best(bs, bi);
    __syncthreads();  // previous round's reads of red_* are done
    if (lane == 0) { red_f[wv] = bs; red_i[wv] = bi; }
    __syncthreads();
    bs = red_f[0]; bi = red_i[0];
#pragma unroll
    for (int k = 1; k < BEAM_THREADS / 64; ++k)
      if (better(red_f[k], red_i[k], bs, bi)) { bs = red_f[k]; bi = red_i[k]; }
    ps = bs; pi = bi;
    if (tid == 0) { cs[t] = bs; ci[t] = (bi == 0x7fffffff) ? t : bi; }  // (no finite score left, e.g. NaN logits: a valid index)
  }
}

// 1024 threads per row (round 3): 320 rows on 256 CUs is two rounds of whatever ONE workgroup takes, and that is vector-instruction
// time of the per-thread insertion lists -- 117 elements per thread at 256 threads, 29 at 1024 (55 -> ~25 us per step).
constexpr int FAST_THREADS = 1024;
// Same result in TWO passes over the row instead of 2 + beam (297 -> ~40 us per step at 320 rows x 30000): pass A is
// an online (max, sum-exp) with 16-byte loads, pass B keeps each thread's own K best (score desc, index asc) in
// registers -- a thread visits indices in increasing order, so strict '>' insertion keeps the lowest index first among
// equal scores -- and K rounds of a block arg-best over the threads' list heads pick the row's K best in order.
template <int K>
__global__ __launch_bounds__(FAST_THREADS) void beam_row_topk_fast_kernel(imt_beam_args a) {
  __shared__ float red_f[FAST_THREADS / 64];
  __shared__ float red_g[FAST_THREADS / 64];
  __shared__ int red_i[FAST_THREADS / 64];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = r / a.rep;
  const float* x = a.logits + (int64_t)r * a.ld;
  const float cur = a.scores_in[r];
  const bool use_pen = a.beam > 1;
  const float pen = use_pen ? powf((a.sizes_in[r] + 6.0f) / 6.0f, a.len_penalty_ratio) : 1.0f;
  const bool masked = a.eos_in[r] || (a.step > 1 && a.max_lens[b] < (int64_t)a.step + 1);
  float* cs = a.cand_scores + (int64_t)r * a.beam;
  int* ci = a.cand_idx + (int64_t)r * a.beam;
  if (masked) {
    const float s = use_pen ? (cur + 0.0f) / pen : cur + 0.0f;
    for (int t = tid; t < a.beam; t += FAST_THREADS) { cs[t] = s; ci[t] = t; }
    return;
  }
  const int V4 = a.V & ~3;
  // pass A: online max / sum-exp
  // NS sweeps of the row are REQUESTED together before the first is consumed (clamped addresses, the value masked where used): one
  // sweep per loop trip was a chain of ~8 exposed load latencies per pass -- 41 us per row where the row's 120 KB take ~1 us to stream
  constexpr int NS = 4;
  float m = -INFINITY, l = 0.f;
  for (int v0 = tid * 4; v0 < V4; v0 += NS * FAST_THREADS * 4) {
    f32x4 qs[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qs[s] = *reinterpret_cast<const f32x4*>(x + min(v0 + s * FAST_THREADS * 4, V4 - 4));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (v0 + s * FAST_THREADS * 4 < V4) {
        const f32x4 q = qs[s];
        const float qm = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3]));
        if (qm > m) { l *= __expf(m - qm); m = qm; }
        l += __expf(q[0] - m) + __expf(q[1] - m) + __expf(q[2] - m) + __expf(q[3] - m);
      }
    }
  }
  for (int v = V4 + tid; v < a.V; v += FAST_THREADS) {
    const float q = x[v];
    if (q > m) { l *= __expf(m - q); m = q; }
    l += __expf(q - m);
  }
  float mx = wave_max(m);
  if (lane == 0) red_f[wv] = mx;
  __syncthreads();
  mx = red_f[0];
#pragma unroll
  for (int k = 1; k < FAST_THREADS / 64; ++k) mx = fmaxf(mx, red_f[k]);
  float sum = wave_sum(m == -INFINITY ? 0.f : l * __expf(m - mx));
  if (lane == 0) red_g[wv] = sum;
  __syncthreads();
  float tot = red_g[0];
#pragma unroll
  for (int k = 1; k < FAST_THREADS / 64; ++k) tot += red_g[k];
  const float lse = mx + logf(tot);
  // pass B: per-thread top-K
  float ls[K]; int li[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { ls[k] = -INFINITY; li[k] = 0x7fffffff; }
  auto offer = [&](float q, int v) {
    const float lp = q - lse;
    const float sc = use_pen ? (cur + lp) / pen : cur + lp;
    if (sc > ls[K - 1]) {  // strictly better than this thread's current worst (an equal score keeps the earlier index)
      bool placed = false;
#pragma unroll
      for (int k = K - 1; k > 0; --k) {
        if (!placed) {
          if (sc > ls[k - 1]) { ls[k] = ls[k - 1]; li[k] = li[k - 1]; }  // slot k-1 moves down, keep looking
          else { ls[k] = sc; li[k] = v; placed = true; }
        }
      }
      if (!placed) { ls[0] = sc; li[0] = v; }
    }
  };
  for (int v0 = tid * 4; v0 < V4; v0 += NS * FAST_THREADS * 4) {
    f32x4 qs[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) qs[s] = *reinterpret_cast<const f32x4*>(x + min(v0 + s * FAST_THREADS * 4, V4 - 4));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int v = v0 + s * FAST_THREADS * 4;
      if (v < V4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) offer(qs[s][e], v + e);
      }
    }
  }
  for (int v = V4 + tid; v < a.V; v += FAST_THREADS) offer(x[v], v);
  // The row's K best in two stages without a block-wide round per pick (each cost ~2.4 us on 16 waves: two barriers and a scan of 16
  // LDS slots): (1) every wave picks ITS K best from its lanes' list heads -- butterfly arg-best, the owning lane pops -- and parks them;
  // (2) one barrier, then wave 0 holds one parked list per lane and picks the row's K best the same way.  Same order as before:
  // (score desc, index asc) at every comparison, indices are unique.
  __shared__ float wl_s[FAST_THREADS / 64][K];
  __shared__ int wl_i[FAST_THREADS / 64][K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    if (t < a.beam) {
      float bs = ls[0]; int bi = li[0];
      wave_best(bs, bi);
      if (li[0] == bi && bi != 0x7fffffff) {  // this lane owned the winner: pop it
#pragma unroll
        for (int k = 0; k < K - 1; ++k) { ls[k] = ls[k + 1]; li[k] = li[k + 1]; }
        ls[K - 1] = -INFINITY; li[K - 1] = 0x7fffffff;
      }
      if (lane == 0) { wl_s[wv][t] = bs; wl_i[wv][t] = bi; }
    }
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const bool have = lane < FAST_THREADS / 64 && k < a.beam;
      ls[k] = have ? wl_s[lane][k] : -INFINITY;
      li[k] = have ? wl_i[lane][k] : 0x7fffffff;
    }
#pragma unroll
    for (int t = 0; t < K; ++t) {
      if (t < a.beam) {
        float bs = ls[0]; int bi = li[0];
        wave_best(bs, bi);
        if (li[0] == bi && bi != 0x7fffffff) {
#pragma unroll
          for (int k = 0; k < K - 1; ++k) { ls[k] = ls[k + 1]; li[k] = li[k + 1]; }
          ls[K - 1] = -INFINITY; li[K - 1] = 0x7fffffff;
        }
        if (lane == 0) { cs[t] = bs; ci[t] = (bi == 0x7fffffff) ? t : bi; }  // (no finite score left, e.g. NaN logits: a valid index)
      }
    }
  }
}

// Stage 2: one wave per sentence merges rep*beam candidates and does the bookkeeping.
__global__ __launch_bounds__(64) void beam_merge_kernel(imt_beam_args a) {
  __shared__ float top_s[MAX_BEAM];
  __shared__ long long top_f[MAX_BEAM];
  __shared__ int s_parent[MAX_BEAM];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int ncand = a.rep * a.beam;
  const float* cs = a.cand_scores + (int64_t)b * ncand;
  const int* ci = a.cand_idx + (int64_t)b * ncand;
  // selection: candidate c (row c / beam, rank c % beam) has flat index (c / beam) * V + idx; candidates of one row
  // are already ordered, so "lowest flat index among equal scores" == lowest c among equal scores... only within a
  // row; across rows the row order decides -- also lowest c.  Hence rank by (score desc, c asc).
  float ps = INFINITY; int pc = -1;
  for (int t = 0; t < a.beam; ++t) {
    float bs = -INFINITY; int bc = 0x7fffffff;
    for (int c = lane; c < ncand; c += 64) {
      const float s = cs[c];
      const bool eligible = (t == 0) || s < ps || (s == ps && c > pc);
      if (eligible && better(s, c, bs, bc)) { bs = s; bc = c; }
    }
    wave_best(bs, bc);
    // NaN scores (bf16 overflow, a damaged checkpoint) are never eligible: fall back to candidate t so that every index
    // below stays inside its buffer -- the hypothesis is garbage either way, the access must not be
    if (bc == 0x7fffffff) bc = t;
    ps = bs; pc = bc;
    if (lane == 0) {
      int word = ci[bc];
      if ((unsigned)word >= (unsigned)a.V) word = 0;
      top_s[t] = bs; top_f[t] = (long long)(bc / a.beam) * a.V + word;
    }
  }
  __syncthreads();
  const bool over = a.step > 1 && a.max_lens[b] < (int64_t)a.step + 1;
  if (lane < a.beam) {
    const int t = lane;
    long long f = top_f[t];
    if (a.step > 1) {
      if (over) f = a.pad_idx;                            // :205-207
      if (a.eos_in[(int64_t)b * a.rep + t]) f = a.pad_idx;  // :211-212 (old row's flag applied to the new slot)
    }
    const int parent = (a.step > 1) ? (int)(f / a.V) : 0;  // :216 (floor division)
    const long long word = f % a.V;
    const int prow = b * a.rep + parent;
    const int orow = b * a.beam + t;
    s_parent[t] = prow;
    a.scores_out[orow] = top_s[t];
    if (a.beam > 1) a.sizes_out[orow] = a.sizes_in[prow] + (word != a.pad_idx ? 1.0f : 0.0f);
    const bool has_eos = a.eos_in[prow] || word == a.eos;
    a.eos_out[orow] = has_eos ? 1 : 0;
    a.parent_out[orow] = prow;
    a.tokens_out[orow] = word;
    a.hist_out[(int64_t)orow * a.t_max + a.step] = word;
    if (a.slots_out && a.step < a.t_max) a.slots_out[(int64_t)orow * a.t_max + a.step] = orow;
    if (a.eos_count && has_eos) atomicAdd(a.eos_count + a.step, 1);
  }
  __syncthreads();
  for (int t = 0; t < a.beam; ++t) {
    const int prow = s_parent[t], orow = b * a.beam + t;
    for (int j = lane; j < a.step; j += 64) {
      a.hist_out[(int64_t)orow * a.t_max + j] = a.hist_in[(int64_t)prow * a.t_max + j];
      if (a.slots_out) a.slots_out[(int64_t)orow * a.t_max + j] = a.slots_in[(int64_t)prow * a.t_max + j];
    }
  }
}

}  // namespace

extern "C" int imt_attention_decode(const imt_attn_decode_args* a, void* stream) {
  IMT_CHECK_ARG(a, "attention_decode: null args");
  IMT_CHECK_ARG(a->dtype == IMT_F32 || a->dtype == IMT_BF16, "attention_decode: bad dtype");
  IMT_CHECK_ARG(a->head_dim == 32 || a->head_dim == 64, "attention_decode: head_dim %d unsupported (32 or 64)", a->head_dim);
  IMT_CHECK_ARG(a->R > 0 && a->H > 0 && a->n_keys > 0 && a->rep > 0 && a->R % a->rep == 0, "attention_decode: bad sizes");
  IMT_CHECK_ARG(a->Q && a->K && a->V && a->O, "attention_decode: null tensor");
  IMT_CHECK_ARG(a->ldq % 8 == 0 && a->ld_row % 8 == 0 && a->ld_pos % 8 == 0 && a->ldo % 8 == 0, "attention_decode: strides must be multiples of 8 elements");
  IMT_CHECK_ARG(!a->slots || a->ld_slots >= a->n_keys, "attention_decode: slot table narrower than n_keys");
  IMT_CHECK_ARG(!a->key_mask || a->ld_mask >= a->n_keys, "attention_decode: key mask narrower than n_keys");
  hipStream_t st = (hipStream_t)stream;
  const int waves = a->R * a->H;
  const dim3 grid(imt_cdiv(waves, 4)), block(256);
  const double es = a->dtype == IMT_BF16 ? 2 : 4;
  ImtProfScope prof("attn_decode", 4.0 * waves * a->n_keys * a->head_dim, 2.0 * waves * a->n_keys * a->head_dim * es, st);
  if (a->dtype == IMT_BF16) {
    if (a->head_dim == 64) hipLaunchKernelGGL((attn_decode_kernel<bf16_t, 64>), grid, block, 0, st, *a);
    else hipLaunchKernelGGL((attn_decode_kernel<bf16_t, 32>), grid, block, 0, st, *a);
  } else {
    if (a->head_dim == 64) hipLaunchKernelGGL((attn_decode_kernel<float, 64>), grid, block, 0, st, *a);
    else hipLaunchKernelGGL((attn_decode_kernel<float, 32>), grid, block, 0, st, *a);
  }
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_beam_step(const imt_beam_args* a, void* stream) {
  IMT_CHECK_ARG(a, "beam_step: null args");
  IMT_CHECK_ARG(a->B > 0 && a->beam > 0 && a->beam <= MAX_BEAM && a->V >= a->beam, "beam_step: beam %d out of range (1..%d, <= V)", a->beam, MAX_BEAM);
  IMT_CHECK_ARG(a->rep == 1 || a->rep == a->beam, "beam_step: rep must be 1 (first step) or beam");
  IMT_CHECK_ARG(a->step >= 1 && a->step < a->t_max, "beam_step: step %d outside [1, t_max=%d)", a->step, a->t_max);
  IMT_CHECK_ARG(a->step == 1 || a->rep == a->beam, "beam_step: only the first step may have rep == 1");
  IMT_CHECK_ARG(a->pad_idx >= 0 && a->pad_idx < (int64_t)a->beam * a->V, "beam_step: pad_idx out of range");
  IMT_CHECK_ARG(a->logits && a->scores_in && a->eos_in && a->max_lens && a->hist_in && a->cand_scores && a->cand_idx, "beam_step: null input");
  IMT_CHECK_ARG(a->beam == 1 || (a->sizes_in && a->sizes_out), "beam_step: sizes required for beam > 1");
  IMT_CHECK_ARG(a->scores_out && a->eos_out && a->hist_out && a->parent_out && a->tokens_out, "beam_step: null output");
  IMT_CHECK_ARG((a->slots_in == nullptr) == (a->slots_out == nullptr), "beam_step: slots_in/slots_out must both be given or both NULL");
  hipStream_t st = (hipStream_t)stream;
  const int rows = a->B * a->rep;
  {
    ImtProfScope prof("beam_row_topk", 0, (double)rows * a->V * 4 * (2 + a->beam), st);
    const bool vec_ok = (a->ld % 4 == 0) && (((uintptr_t)a->logits & 15) == 0);
    if (vec_ok && a->beam <= 4) hipLaunchKernelGGL(beam_row_topk_fast_kernel<4>, dim3(rows), dim3(FAST_THREADS), 0, st, *a);
    else if (vec_ok && a->beam <= 8) hipLaunchKernelGGL(beam_row_topk_fast_kernel<8>, dim3(rows), dim3(FAST_THREADS), 0, st, *a);
    else hipLaunchKernelGGL(beam_row_topk_kernel, dim3(rows), dim3(BEAM_THREADS), 0, st, *a);
    IMT_CHECK_LAUNCH();
  }
  {
    ImtProfScope prof("beam_merge", 0, (double)a->B * a->beam * a->step * 12, st);
    hipLaunchKernelGGL(beam_merge_kernel, dim3(a->B), dim3(64), 0, st, *a);
    IMT_CHECK_LAUNCH();
  }
  return IMT_OK;
}
