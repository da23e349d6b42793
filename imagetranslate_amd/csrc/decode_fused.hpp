// Internal interface between the stack runtime (model.hip) and the one-launch decoder step (decode_fused.hip).
#pragma once
#include "common.hpp"

constexpr int IMT_FUSED_MAX_LAYERS = 12;
constexpr int IMT_FUSED_BAR_WORDS = 32 * 20; // barrier counters (own 128-B lines) + status word

struct ImtFusedLayer {
  // parameters (bf16, in the flat store)
  const bf16_t *wqkv, *bqkv, *wo, *bo, *g1, *b1;          // self attention block + its LayerNorm
  const bf16_t *wq, *bq, *wo2, *bo2, *g2, *b2;            // cross attention: query projection, output projection, LayerNorm
  const bf16_t *w1, *bf1, *w2, *bf2, *g3, *b3;            // feed-forward + output LayerNorm
  bf16_t* cache;                                          // self K|V cache of the layer [r_max][t_max][3d]
  const bf16_t* cross_kv;                                 // [B][Tk][2d]
  // hand-off buffers of this layer, each written once per launch (never re-read stale from a cache)
  bf16_t *xin, *ctx1, *a, *q, *ctx2, *b, *h;              // [r_max][d] ... h: [r_max][ff]
  float *pre1, *pre2, *pre3;                              // pre-LayerNorm sums, fp32 [r_max][d]
};

struct ImtFusedArgs {
  ImtFusedLayer L[IMT_FUSED_MAX_LAYERS];
  int n_layers, d, R, rep, pos, Tk, t_max, r_max, H, dh, ff;   // d: 512 or 768
  // BertEmbeddings of the newest tokens, computed while layer 0 stages its operand: word[ids] + pos[pos_ids] + type[type_ids] -> LayerNorm
  const int64_t *ids, *pos_ids, *type_ids;   // [R]; type_ids nullable (zeros)
  const bf16_t *emb_word, *emb_pos, *emb_type, *emb_g, *emb_b;
  int vocab, max_pos, n_types;
  bf16_t* out;             // [R][d]
  const int32_t* slots; const uint8_t* enc_mask;
  float eps;
  unsigned long long* trace;  // tuning only (IMT_DECODE_TRACE): nullptr or [256][128] time stamps
  unsigned* bar;           // IMT_FUSED_BAR_WORDS words, zero at launch; bar[IMT_FUSED_BAR_WORDS - 1] is the status word (sticky)
};

int64_t imt_decode_fused_layer_bytes(int r_max, int d, int ff);   // hand-off buffers of one layer
bool imt_decode_fused_shape(int d, int heads, int ff, int n_layers);   // bf16 stacks the one-launch step is built for
bool imt_decode_fused_enabled();
int imt_decode_fused_launch(const ImtFusedArgs& a, hipStream_t st);
