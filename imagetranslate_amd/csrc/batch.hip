// On-device batch construction for MASS (src/utils.py:41-82): span selection, the shifted
// decoder input with its original positions, compaction of the hidden tokens and the 80/10/10 replacement -- integer
// work, one workgroup per sentence, so a MASS step needs no per-token host loop and no host->device copy of masks.
// Randomness is counter-based (mix32 of (seed, stream, index)), restated bit for bit by oracle/batch_oracle.py.
#include "common.hpp"

namespace {

IMT_DEVICE float mass_uniform(uint64_t seed, uint32_t stream, uint32_t index) {  // [0, 1) with 24 bits
  uint32_t h = mix32(index ^ (uint32_t)seed);
  h = mix32(h + stream * 0x9e3779b9U + (uint32_t)(seed >> 32));
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(256) void mass_mask_kernel(imt_mass_args a) {
  __shared__ int s_first, s_len;
  const int row = blockIdx.x;
  int64_t* text = a.src_text + (int64_t)row * a.width;
  if (threadIdx.x == 0) {
    const int64_t pad = a.pad_indices[row];
    const int span_len = (int)(pad / 2);
    // float32 with TWO roundings like the reference's tensor expression pad - (1-p)*pad: no fused multiply-add (HIP's
    // __fmul_rn / __fsub_rn are plain operators and would be contracted), e.g. pad = 50: 50 - 35.0 = 15, fma gives 15.0000006
    float bound;
    {
#pragma clang fp contract(off)
      const float prod = (1.0f - a.mask_prob) * (float)pad;
      bound = (float)pad - prod;
    }
    const int hint = (int)ceilf(bound);
    const float u0 = mass_uniform(a.seed, 0u, (uint32_t)row);
    int first;
    if (u0 > 0.8f) first = 1;
    else if (u0 > 0.6f) first = hint;
    else if (hint >= 2) {
      first = 2 + (int)(mass_uniform(a.seed, 1u, (uint32_t)row) * (float)(hint - 1));
      if (first > hint) first = hint;
    } else first = 2;
    s_first = first; s_len = span_len;
  }
  __syncthreads();
  const int first = s_first, len = s_len;
  const int last = min(first + len, a.width);  // python slice semantics
  const int64_t off = a.row_offsets[row];
  // span mask
  for (int c = threadIdx.x; c < a.width; c += 256) a.src_mask[(int64_t)row * a.width + c] = (c >= first && c < last) ? 1 : 0;
  // decoder input = tokens first-1 .. last-1 with their original positions (read BEFORE the replacement below)
  for (int k = threadIdx.x; k < a.recover_width; k += 256) {
    const int c = first - 1 + k;
    const bool in = (c >= 0 && c < last && k <= last - first);
    a.to_recover[(int64_t)row * a.recover_width + k] = in ? text[c] : a.pad_id;
    a.positions[(int64_t)row * a.recover_width + k] = in ? (int64_t)c : (int64_t)(a.width - 1);
  }
  __syncthreads();
  // hidden tokens: compacted originals (== prediction targets) and the 80/10/10 replacement, in place
  for (int k = threadIdx.x; k < last - first; k += 256) {
    const int c = first + k;
    const int64_t orig = text[c];
    a.targets[off + k] = orig;
    const uint32_t idx = (uint32_t)(row * a.width + c);
    const float u = mass_uniform(a.seed, 2u, idx);
    int64_t repl = orig;
    if (u < 0.8f) repl = a.mask_id;
    else if (u < 0.9f) {
      const int span = a.vocab - a.n_special;
      int r = (int)(mass_uniform(a.seed, 3u, idx) * (float)span);
      if (r >= span) r = span - 1;
      repl = a.n_special + r;
    }
    text[c] = repl;
  }
}

__global__ __launch_bounds__(256) void mass_unmask_kernel(int64_t* text, const uint8_t* mask, const int64_t* originals,
                                                          const int64_t* row_offsets, int width) {
  __shared__ int s_first;
  const int row = blockIdx.x;
  if (threadIdx.x == 0) s_first = width;
  __syncthreads();
  int mine = width;
  for (int c = threadIdx.x; c < width; c += 256)
    if (mask[(int64_t)row * width + c]) { mine = c; break; }
  atomicMin(&s_first, mine);
  __syncthreads();
  const int first = s_first;
  for (int c = first + threadIdx.x; c < width; c += 256)
    if (mask[(int64_t)row * width + c]) text[(int64_t)row * width + c] = originals[row_offsets[row] + (c - first)];
}

// Non-pad target selection (src/seq2seq.py:175-177, train_image_mt.py:253-256): flat positions p = b*T1 + t of the
// positions whose mask[b, col0 + t] is set, in order, plus the target ids at those positions and their count.  One
// workgroup walks the B*T1 flags in chunks of 1024 with a running offset (ballot + popcount scan per wave, 16 wave
// totals through LDS): ~5 us at 8128 positions instead of ~90 us of nonzero / boolean-index kernels.
// One workgroup, ordered compaction.  Thread t owns the SEL_ITEMS consecutive positions of chunk-local index t; all its
// mask bytes and ids are requested together, unconditionally (clamped), so a chunk of 16384 positions costs two memory
// round trips and one block-wide exclusive scan (the first version walked 1024 positions per trip with three barriers
// each: 60 us for the 8128 positions of a C1 batch, on the critical path in front of the encoder; this one ~6 us).
constexpr int SEL_ITEMS = 16;
__global__ __launch_bounds__(1024) void select_plan_kernel(const uint8_t* __restrict__ mask, int64_t ld_mask,
                                                           const int64_t* __restrict__ ids, int64_t ld_ids, int B, int T1, int col0,
                                                           int32_t* __restrict__ idx, int64_t* __restrict__ targets,
                                                           int32_t* __restrict__ count) {
  __shared__ int wave_tot[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = B * T1;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int p0 = 0; p0 < n; p0 += 1024 * SEL_ITEMS) {
    uint8_t mk[SEL_ITEMS];
    int64_t tok[SEL_ITEMS];
    const int first = p0 + tid * SEL_ITEMS;
#pragma unroll
    for (int k = 0; k < SEL_ITEMS; ++k) {
      const int p = min(first + k, n - 1);
      const int b = p / T1, t = p - b * T1;
      mk[k] = mask[(int64_t)b * ld_mask + col0 + t];
      tok[k] = ids[(int64_t)b * ld_ids + col0 + t];
    }
    int mine = 0;
#pragma unroll
    for (int k = 0; k < SEL_ITEMS; ++k) mine += (first + k < n && mk[k] != 0) ? 1 : 0;
    // exclusive scan of `mine` over the workgroup: inclusive wave scan by shuffles, wave totals through LDS
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int off = s_base + incl - mine;
    for (int k = 0; k < wv; ++k) off += wave_tot[k];
#pragma unroll
    for (int k = 0; k < SEL_ITEMS; ++k) {
      if (first + k < n && mk[k] != 0) {
        idx[off] = first + k;
        targets[off] = tok[k];
        ++off;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int tot = s_base;
      for (int k = 0; k < 16; ++k) tot += wave_tot[k];
      s_base = tot;
    }
    __syncthreads();
  }
  if (tid == 0) count[0] = s_base;
}

}  // namespace

extern "C" int imt_select_plan(const uint8_t* mask, int64_t ld_mask, const int64_t* ids, int64_t ld_ids, int B, int T1, int col0,
                               int32_t* idx, int64_t* targets, int32_t* count, void* stream) {
  IMT_CHECK_ARG(mask && ids && idx && targets && count && B > 0 && T1 >= 0 && col0 >= 0, "select_plan: bad args");
  IMT_CHECK_ARG((int64_t)B * T1 < (1ll << 31), "select_plan: too many positions");
  hipLaunchKernelGGL(select_plan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, ld_mask, ids, ld_ids, B, T1, col0, idx,
                     targets, count);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_mass_mask(const imt_mass_args* a, void* stream) {
  IMT_CHECK_ARG(a, "mass_mask: null args");
  IMT_CHECK_ARG(a->n_rows > 0 && a->width > 1 && a->recover_width > 0, "mass_mask: bad sizes");
  IMT_CHECK_ARG(a->mask_prob > 0.f && a->mask_prob < 1.f, "mass_mask: mask_prob must be in (0, 1)");
  IMT_CHECK_ARG(a->vocab > a->n_special && a->n_special >= 0, "mass_mask: vocabulary smaller than the special-token block");
  IMT_CHECK_ARG(a->src_text && a->pad_indices && a->row_offsets && a->src_mask && a->to_recover && a->positions && a->targets,
                "mass_mask: null tensor");
  hipStream_t st = (hipStream_t)stream;
  ImtProfScope prof("mass_mask", 0, (double)a->n_rows * a->width * 17, st);
  hipLaunchKernelGGL(mass_mask_kernel, dim3(a->n_rows), dim3(256), 0, st, *a);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}

extern "C" int imt_mass_unmask(int64_t* src_text, const uint8_t* src_mask, const int64_t* originals, const int64_t* row_offsets,
                               int n_rows, int width, void* stream) {
  IMT_CHECK_ARG(src_text && src_mask && originals && row_offsets && n_rows > 0 && width > 0, "mass_unmask: bad args");
  hipLaunchKernelGGL(mass_unmask_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, src_text, src_mask, originals, row_offsets, width);
  IMT_CHECK_LAUNCH();
  return IMT_OK;
}
