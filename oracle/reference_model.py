"""CPU oracle: fp32 plain-PyTorch restatement of ImageTranslate's encoder-decoder hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Each class cites the
reference file:line it follows (paths relative to the reference repo root).  The
arithmetic of the transformer blocks lives in the reference's un-vendored
dependency ``transformers==2.9.0`` (``src/requirements.txt:8``; modules
``modeling_bert`` / ``modeling_utils``); its published algorithm is restated here:

  * embeddings  : LN(word[ids] + pos[position_ids or arange] + type[token_type_ids]), eps 1e-12,
                  then dropout(hidden_dropout_prob); word table has padding_idx = pad id.
  * attention   : Q,K,V = Linear(x); scores = Q K^T / sqrt(d_h) + additive_mask; softmax(-1);
                  dropout(attention_probs_dropout_prob); ctx = P V; merge heads.
                  Cross-attention takes K,V from encoder states and the encoder mask.
  * masks       : 2-D mask -> [B,1,1,S]; 3-D -> [B,1,T,T]; 2-D with is_decoder -> AND causal;
                  additive = (1 - m) * -10000.0
  * layer       : post-LN.  a = LN(dropout(Wo ctx) + x); [cross: a = LN(dropout(Wo' ctx') + a)];
                  h = gelu_erf(W1 a); y = LN(dropout(W2 h) + a)
  * init        : Linear / Embedding weights N(0, 0.02); biases 0; LN weight 1, bias 0.

PARITY UNPINNED at the whole-model level in the strict sense (the reference's tests hold
only an output shape assertion, ``src/tests/test_model.py:70-74``, and 2.9.0 cannot be
installed).  What IS pinned (``tests/test_oracle_pinning.py``): ``SmoothedNLLLoss`` against
the reference's own ``src/loss.py``; the leaf blocks, the whole ``BertEncoder`` stack
(encoder and cross-attending decoder mode, forward + every gradient) and the
``BertEncoderModel`` / ``BertDecoderModel`` compositions against the installed
``transformers`` (5.15) ``BertEncoder`` / ``BertModel`` on the same state dict (identical key
sets), with the reference's mask forms.

Deliberate, documented deviation: ``num_attention_heads`` is a knob (reference
hard-codes 12, ``src/lm_config.py:13``) so that BASELINE's d=512/h=8 and d=128/h=4
configurations are constructible.
"""
import copy
import math
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- config
def bert_config(vocab_size: int, pad_token_id: int, bos_token_id: int, eos_token_id: int,
                enc_layer: int = 6, embed_dim: int = 768, intermediate_dim: int = 3072,
                num_attention_heads: int = 12) -> Dict:
    """src/lm_config.py:4-30."""
    return {
        "attention_probs_dropout_prob": 0.1,
        "hidden_act": "gelu",
        "hidden_dropout_prob": 0.1,
        "hidden_size": embed_dim,
        "initializer_range": 0.02,
        "intermediate_size": intermediate_dim,
        "max_position_embeddings": 512,
        "num_attention_heads": num_attention_heads,
        "num_hidden_layers": enc_layer,
        "vocab_size": vocab_size,
        "pad_token_id": pad_token_id,
        "bos_token_id": bos_token_id,
        "eos_token_id": eos_token_id,
    }


class BertConfig:
    """Attribute bag standing in for transformers.BertConfig (src/seq2seq.py:37)."""

    def __init__(self, **kw):
        self.layer_norm_eps = 1e-12
        self.type_vocab_size = 2
        self.is_decoder = False
        for k, v in kw.items():
            setattr(self, k, v)
        if self.hidden_size % self.num_attention_heads != 0:
            raise ValueError("hidden size %d is not a multiple of the number of attention heads %d"
                             % (self.hidden_size, self.num_attention_heads))


class SyntheticTextProcessor:
    """Duck-typed stand-in for src/textprocessor.py:10-206 (ids only; no tokenizer).

    Special ids follow src/textprocessor.py:22-31: pad=0, <s>=1, <unk>=2, <mask>=3, </s>=4,
    then one id per language tag.
    """

    class _Tok:
        def __init__(self, v):
            self._v = v

        def get_vocab_size(self):
            return self._v

    def __init__(self, vocab_size: int = 1000, languages: Optional[Dict[str, int]] = None):
        self.languages = languages if languages is not None else {"<en>": 0, "<fa>": 1}
        self.tokenizer = self._Tok(vocab_size)
        self.special_tokens = ["<pad>", "<s>", "<unk>", "<mask>", "</s>"] + list(self.languages.keys())

    def pad_token_id(self): return 0
    def bos_token_id(self): return 1
    def unk_token_id(self): return 2
    def mask_token_id(self): return 3
    def sep_token_id(self): return 4
    def vocab_size(self): return self.tokenizer.get_vocab_size()

    def token_id(self, tok):
        return self.special_tokens.index(tok) if tok in self.special_tokens else 0

    def lang_id(self, tok):
        return self.languages.get(tok, 0)


# --------------------------------------------------------------------------- HF-BERT 2.9.0 blocks
class BertEmbeddings(nn.Module):
    """transformers 2.9.0 modeling_bert.BertEmbeddings (instantiated at src/bert_seq2seq.py:21,99)."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None):
        input_shape = input_ids.size()
        seq_length = input_shape[1]
        if position_ids is None:
            position_ids = torch.arange(seq_length, dtype=torch.long, device=input_ids.device)
            position_ids = position_ids.unsqueeze(0).expand(input_shape)
        if token_type_ids is None:
            token_type_ids = torch.zeros(input_shape, dtype=torch.long, device=input_ids.device)
        x = self.word_embeddings(input_ids) + self.position_embeddings(position_ids) \
            + self.token_type_embeddings(token_type_ids)
        return self.dropout(self.LayerNorm(x))


class BertSelfAttention(nn.Module):
    """transformers 2.9.0 modeling_bert.BertSelfAttention."""

    def __init__(self, config):
        super().__init__()
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def _split(self, x):
        return x.view(x.size(0), x.size(1), self.num_attention_heads, self.attention_head_size).permute(0, 2, 1, 3)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None):
        q = self.query(hidden_states)
        if encoder_hidden_states is not None:
            k = self.key(encoder_hidden_states)
            v = self.value(encoder_hidden_states)
            attention_mask = encoder_attention_mask
        else:
            k = self.key(hidden_states)
            v = self.value(hidden_states)
        q, k, v = self._split(q), self._split(k), self._split(v)
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self.attention_head_size)
        if attention_mask is not None:
            scores = scores + attention_mask
        probs = self.dropout(F.softmax(scores, dim=-1))
        ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous()
        return ctx.view(ctx.size(0), ctx.size(1), self.all_head_size)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dropout(self.dense(hidden_states)) + input_tensor)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None):
        ctx = self.self(hidden_states, attention_mask, encoder_hidden_states, encoder_attention_mask)
        return self.output(ctx, hidden_states)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)

    def forward(self, hidden_states):
        return F.gelu(self.dense(hidden_states))  # exact erf GELU (src/lm_config.py:7)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, hidden_states, input_tensor):
        return self.LayerNorm(self.dropout(self.dense(hidden_states)) + input_tensor)


class BertLayer(nn.Module):
    """transformers 2.9.0 modeling_bert.BertLayer: every is_decoder layer owns a crossattention block."""

    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.is_decoder = config.is_decoder
        if self.is_decoder:
            self.crossattention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None):
        a = self.attention(hidden_states, attention_mask)
        if self.is_decoder and encoder_hidden_states is not None:
            a = self.crossattention(a, attention_mask, encoder_hidden_states, encoder_attention_mask)
        return self.output(self.intermediate(a), a)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states, attention_mask=None, encoder_hidden_states=None, encoder_attention_mask=None):
        for layer in self.layer:
            hidden_states = layer(hidden_states, attention_mask, encoder_hidden_states, encoder_attention_mask)
        return hidden_states


def _init_bert_weights(module, std):
    """transformers 2.9.0 BertPreTrainedModel._init_weights."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=std)
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()


def extended_attention_mask(attention_mask, is_decoder: bool, dtype=torch.float32):
    """transformers 2.9.0 modeling_utils.get_extended_attention_mask (src/bert_seq2seq.py:69-71,130-132)."""
    if attention_mask.dim() == 3:
        ext = attention_mask[:, None, :, :]
    elif attention_mask.dim() == 2:
        if is_decoder:
            b, s = attention_mask.shape
            ids = torch.arange(s, device=attention_mask.device)
            causal = ids[None, None, :].repeat(b, s, 1) <= ids[None, :, None]
            causal = causal.to(attention_mask.dtype)
            ext = causal[:, None, :, :] * attention_mask[:, None, None, :]
        else:
            ext = attention_mask[:, None, None, :]
    else:
        raise ValueError("Wrong shape for attention_mask")
    ext = ext.to(dtype=dtype)
    return (1.0 - ext) * -10000.0


class BertOutputLayer(nn.Module):
    """src/bert_seq2seq.py:6-12."""

    def __init__(self, config):
        super().__init__()
        self.layer = nn.Linear(config.hidden_size, config.vocab_size)

    def forward(self, input):
        return self.layer(input)


class _Pretrained(nn.Module):
    def init_weights(self):
        std = self.config.initializer_range
        self.apply(lambda m: _init_bert_weights(m, std))

    @staticmethod
    def _tie_or_clone_weights(output_embeddings, input_embeddings):
        """transformers 2.9.0 modeling_utils._tie_or_clone_weights (first arg receives the weight)."""
        output_embeddings.weight = input_embeddings.weight
        if getattr(output_embeddings, "bias", None) is not None:
            pass
        if hasattr(output_embeddings, "out_features") and hasattr(input_embeddings, "num_embeddings"):
            output_embeddings.out_features = input_embeddings.num_embeddings


class BertEncoderModel(_Pretrained):
    """src/bert_seq2seq.py:94-144."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.init_weights()

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None):
        input_shape = input_ids.size()
        device = input_ids.device
        if attention_mask is None:
            attention_mask = torch.ones(input_shape, device=device)
        if token_type_ids is None:
            token_type_ids = torch.zeros(input_shape, dtype=torch.long, device=device)
        ext = extended_attention_mask(attention_mask, self.config.is_decoder)
        x = self.embeddings(input_ids=input_ids, position_ids=position_ids, token_type_ids=token_type_ids)
        return self.encoder(x, attention_mask=ext)


class BertDecoderModel(_Pretrained):
    """src/bert_seq2seq.py:15-91."""

    def __init__(self, config):
        super().__init__()
        self.config = copy.deepcopy(config)
        self.config.is_decoder = True
        self.embeddings = BertEmbeddings(self.config)
        self.decoder = BertEncoder(self.config)
        self.init_weights()

    def forward(self, input_ids=None, encoder_attention_mask=None, tgt_attention_mask=None, token_type_ids=None,
                position_ids=None, encoder_states=None):
        input_shape = input_ids.size()
        device = input_ids.device
        if tgt_attention_mask is None:
            tgt_attention_mask = torch.ones(input_shape, device=device)
        if token_type_ids is None:
            token_type_ids = torch.zeros(input_shape, dtype=torch.long, device=device)
        ext = extended_attention_mask(tgt_attention_mask, True)
        eb, es, _ = encoder_states.size()
        if encoder_attention_mask is None:
            encoder_attention_mask = torch.ones((eb, es), device=device)
        # invert_attention_mask, src/bert_seq2seq.py:25-38
        if encoder_attention_mask.dim() == 3:
            enc_ext = encoder_attention_mask[:, None, :, :]
        else:
            enc_ext = encoder_attention_mask[:, None, None, :]
        enc_ext = (1.0 - enc_ext.to(torch.float32)) * -10000.0
        x = self.embeddings(input_ids=input_ids, position_ids=position_ids, token_type_ids=token_type_ids)
        return self.decoder(x, attention_mask=ext, encoder_hidden_states=encoder_states,
                            encoder_attention_mask=enc_ext)


# --------------------------------------------------------------------------- model API
def future_mask(tgt_mask):
    """src/seq2seq.py:14-17: mask[b,i,j] = (j <= i) & tgt_mask[b,i]  (masks by QUERY row)."""
    attn_shape = (tgt_mask.size(0), tgt_mask.size(1), tgt_mask.size(1))
    fm = torch.triu(torch.ones(attn_shape), diagonal=1).type_as(tgt_mask)
    return ~fm & tgt_mask.unsqueeze(-1)


class Seq2Seq(nn.Module):
    """src/seq2seq.py:20-181."""

    def __init__(self, text_processor, lang_dec: bool = True, use_proposals=False, tie_embed=False,
                 enc_layer: int = 6, dec_layer: int = 3, embed_dim: int = 768, intermediate_dim: int = 3072,
                 freeze_image: bool = False, resnet_depth: int = 1, use_obj: bool = False, *,
                 num_attention_heads: int = 12):
        super().__init__()
        self.text_processor = text_processor
        cfg = bert_config(vocab_size=text_processor.tokenizer.get_vocab_size(),
                          pad_token_id=text_processor.pad_token_id(),
                          bos_token_id=text_processor.bos_token_id(),
                          eos_token_id=text_processor.sep_token_id(),
                          enc_layer=enc_layer, embed_dim=embed_dim, intermediate_dim=intermediate_dim,
                          num_attention_heads=num_attention_heads)
        self.enc_layer, self.dec_layer = enc_layer, dec_layer
        self.embed_dim, self.intermediate_dim = embed_dim, intermediate_dim
        cfg["type_vocab_size"] = len(text_processor.languages)
        self.config = BertConfig(**cfg)
        dec_config = copy.deepcopy(self.config)
        dec_config.num_hidden_layers = self.dec_layer

        self.encoder = BertEncoderModel(self.config)
        self.encoder.init_weights()
        self.lang_dec = lang_dec
        self.tie_embed = tie_embed
        tie = _Pretrained._tie_or_clone_weights
        if not lang_dec:
            self.decoder = BertDecoderModel(dec_config)
            tie(self.encoder.embeddings.position_embeddings, self.decoder.embeddings.position_embeddings)
            tie(self.encoder.embeddings.token_type_embeddings, self.decoder.embeddings.token_type_embeddings)
            tie(self.encoder.embeddings.word_embeddings, self.decoder.embeddings.word_embeddings)
            if tie_embed:
                self.output_layer = BertOutputLayer(dec_config)
                tie(self.output_layer, self.encoder.embeddings.word_embeddings)
                tie(self.encoder.embeddings.position_embeddings, self.decoder.embeddings.position_embeddings)
                tie(self.output_layer, self.decoder.embeddings.word_embeddings)
            else:
                self.output_layer = nn.ModuleList([BertOutputLayer(dec_config) for _ in text_processor.languages])
            if len(self.encoder.encoder.layer) == len(self.decoder.decoder.layer):
                for i in range(len(self.encoder.encoder.layer)):
                    self.decoder.decoder.layer[i].attention = self.encoder.encoder.layer[i].attention
        else:
            dec = BertDecoderModel(dec_config)
            self.decoder = nn.ModuleList([copy.deepcopy(dec) for _ in text_processor.languages])
            self.output_layer = nn.ModuleList([BertOutputLayer(dec_config) for _ in text_processor.languages])
            for i, dec in enumerate(self.decoder):
                if tie_embed:
                    tie(self.output_layer[i], self.encoder.embeddings.word_embeddings)
                    dec.embeddings.position_embeddings = self.encoder.embeddings.position_embeddings
                tie(self.output_layer[i], dec.embeddings.word_embeddings)
                tie(self.encoder.embeddings.token_type_embeddings, dec.embeddings.token_type_embeddings)
        self.use_proposals = use_proposals
        if self.use_proposals:  # src/seq2seq.py:79-83
            self.proposal_embedding = self.encoder.embeddings.word_embeddings
            self.lexical_gate = nn.Parameter(torch.full((1, self.config.hidden_size), 0.1))
            self.lexical_layer_norm = nn.LayerNorm(self.config.hidden_size, eps=self.config.layer_norm_eps)
        self.freeze_image = freeze_image
        self.resnet_depth = resnet_depth

    def encode(self, src_inputs, src_mask, src_langs, images=None):
        return (self.encoder(src_inputs, attention_mask=src_mask, token_type_ids=src_langs), None)

    def attend_proposal(self, decoder_output, proposals, pad_idx):
        """src/seq2seq.py:110-144 for a [B, T, d] decoder output and [B, P] proposal ids: every target position attends
        over the P proposal embeddings (scores = dot products, plain softmax -- the reference's -10000 fill at :132 acts on
        a copy made by boolean indexing and changes nothing, so pad proposals DO take part), rows whose proposals are all
        pad get the constant 1e-8 (:139-140), then sigmoid-gated mix with the decoder output (:142-144) and LayerNorm."""
        emb = self.proposal_embedding(proposals)                                   # [B, P, d]
        scores = torch.einsum("btd,bpd->btp", decoder_output, emb)                 # :129
        probs = torch.softmax(scores, dim=-1)                                      # :133
        values = torch.einsum("btp,bpd->btd", probs, emb)                          # :135
        all_pad = (proposals == pad_idx).all(dim=-1)                               # :138
        values = torch.where(all_pad[:, None, None], torch.full_like(values, 1e-8), values)
        gate = torch.sigmoid(self.lexical_gate + 1e-8)
        return self.lexical_layer_norm(gate * decoder_output + (1 - gate) * values)

    def _decode_and_project(self, encoder_states, enc_mask, tgt_inputs, tgt_mask, tgt_langs_t, batch_lang,
                            position_ids, log_softmax, proposals=None):
        subseq_mask = future_mask(tgt_mask[:, :-1])
        decoder = self.decoder if not self.lang_dec else self.decoder[batch_lang]
        output_layer = self.output_layer if (not self.lang_dec) and self.tie_embed else self.output_layer[batch_lang]
        dec_out = decoder(encoder_states=encoder_states, input_ids=tgt_inputs[:, :-1],
                          encoder_attention_mask=enc_mask, tgt_attention_mask=subseq_mask,
                          position_ids=position_ids, token_type_ids=tgt_langs_t[:, :-1])
        if self.use_proposals:
            dec_out = self.attend_proposal(dec_out, proposals, self.text_processor.pad_token_id())
        flat = dec_out.reshape(-1, dec_out.size(-1))
        sel = flat[tgt_mask[:, 1:].contiguous().view(-1)]
        out = output_layer(sel)
        if log_softmax:
            out = F.log_softmax(out, dim=-1)
        return out

    def forward(self, src_inputs, tgt_inputs, src_mask, tgt_mask, src_langs, tgt_langs, proposals=None,
                log_softmax: bool = False):
        batch_lang = int(tgt_langs[0])
        src_langs_t = src_langs.unsqueeze(-1).expand(-1, src_inputs.size(-1))
        tgt_langs_t = tgt_langs.unsqueeze(-1).expand(-1, tgt_inputs.size(-1))
        encoder_states = self.encode(src_inputs, src_mask, src_langs_t)[0]
        return self._decode_and_project(encoder_states, src_mask, tgt_inputs, tgt_mask, tgt_langs_t, batch_lang,
                                        None, log_softmax, proposals=proposals)


class MassSeq2Seq(Seq2Seq):
    """src/mass_seq2seq.py:6-60."""

    def forward(self, src_inputs, tgt_inputs, src_langs, tgt_langs=None, pad_idx: int = 0, tgt_positions=None,
                log_softmax: bool = False, proposals=None):
        if isinstance(tgt_inputs, list):
            tgt_inputs, src_langs = tgt_inputs[0], src_langs[0]
        if isinstance(src_inputs, list):
            src_inputs = src_inputs[0]
        if isinstance(tgt_positions, list):
            tgt_positions = tgt_positions[0]
        src_pads = src_inputs != pad_idx
        tgt_mask = tgt_inputs != pad_idx
        if tgt_langs is not None:
            return Seq2Seq.forward(self, src_inputs=src_inputs, src_mask=src_pads, tgt_inputs=tgt_inputs,
                                   tgt_mask=tgt_mask, src_langs=src_langs, tgt_langs=tgt_langs,
                                   log_softmax=log_softmax)
        src_langs_t = src_langs.unsqueeze(-1).expand(-1, src_inputs.size(-1))
        batch_lang = int(src_langs[0])
        encoder_states = self.encode(src_inputs, src_pads, src_langs_t)[0]
        tgt_langs_t = src_langs.unsqueeze(-1).expand(-1, tgt_inputs.size(-1))
        pos = tgt_positions[:, :-1] if tgt_positions is not None else None
        return self._decode_and_project(encoder_states, src_pads, tgt_inputs, tgt_mask, tgt_langs_t, batch_lang,
                                        pos, log_softmax)


class ImageHead(nn.Module):
    """The part of ModifiedResnet that is on the path (src/image_model.py:35-41,77-78,107-117):
    frozen region features [B,49,C] -> dropout -> fc (no bias) -> + location_embedding -> dropout."""

    def __init__(self, feat_dim: int, embed_dim: int, dropout: float = 0.1, regions: int = 49):
        super().__init__()
        self.dropout = dropout
        self.fc = nn.Linear(feat_dim, embed_dim, bias=False)
        self.location_embedding = nn.Embedding(regions, embed_dim)

    def forward(self, grid_hidden):
        if self.dropout > 0:
            # NB reference applies this first dropout even in eval mode (F.dropout default training=True,
            # src/image_model.py:37-38); the oracle keeps module-mode semantics and parity runs use dropout=0.
            grid_hidden = F.dropout(grid_hidden, p=self.dropout, training=self.training)
        out = self.fc(grid_hidden) + self.location_embedding.weight.unsqueeze(0)
        if self.dropout > 0 and self.training:
            out = F.dropout(out, p=self.dropout)
        return out, None


class ImageMassSeq2Seq(MassSeq2Seq):
    """src/image_model.py:127-183 -- text branch only (batch is None).  The image trunk is lazy:
    region features enter at ``fc`` (BASELINE config 4 feeds frozen 2048-d region feats)."""

    def __init__(self, text_processor, freeze_image: bool = False, resnet_depth: int = 1, lang_dec: bool = False,
                 use_proposals: bool = False, tie_embed: bool = False, enc_layer: int = 6, dec_layer: int = 3,
                 embed_dim: int = 768, intermediate_dim: int = 3072, use_obj: bool = True, *,
                 num_attention_heads: int = 12, image_feat_dim: int = 2048):
        super().__init__(text_processor=text_processor, tie_embed=tie_embed, lang_dec=lang_dec,
                         use_proposals=use_proposals, enc_layer=enc_layer, dec_layer=dec_layer, embed_dim=embed_dim,
                         intermediate_dim=intermediate_dim, freeze_image=freeze_image, resnet_depth=resnet_depth,
                         num_attention_heads=num_attention_heads)
        self.image_model = ImageHead(image_feat_dim, self.config.hidden_size, self.config.hidden_dropout_prob)
        self.image_model.apply(lambda m: None)
        self.multimodal_attention_gate = nn.Parameter(torch.zeros(1, self.config.hidden_size).fill_(0.1))
        self.image_attention_w = nn.Linear(self.config.hidden_size, 1)
        self.encoder_attention_w = nn.Linear(self.config.hidden_size, 1)

    def forward(self, src_inputs=None, src_pads=None, tgt_inputs=None, src_langs=None, tgt_langs=None,
                pad_idx: int = 0, tgt_positions=None, batch=None, neg_samples=None, neg_mask=None, proposals=None,
                log_softmax: bool = False, **kwargs):
        def un(x):
            return x[0] if isinstance(x, list) else x
        src_inputs, src_pads, tgt_inputs = un(src_inputs), un(src_pads), un(tgt_inputs)
        src_langs, tgt_langs, tgt_positions = un(src_langs), un(tgt_langs), un(tgt_positions)
        if batch is None:
            return MassSeq2Seq.forward(self, src_inputs=src_inputs, tgt_inputs=tgt_inputs, src_langs=src_langs,
                                       tgt_langs=tgt_langs, pad_idx=pad_idx, tgt_positions=tgt_positions,
                                       log_softmax=log_softmax)
        raise NotImplementedError("image+text branch is broken in the reference (SURVEY a16); out of scope")


class ImageCaptioning(ImageMassSeq2Seq):
    """src/image_model.py:267-377 with use_obj=False (--no-obj): image-only decoder path."""

    def __init__(self, *a, **kw):
        kw.setdefault("use_obj", False)
        super().__init__(*a, **kw)

    def encode(self, src_inputs=None, src_mask=None, src_langs=None, images=None):
        if images is not None:
            return self.image_model(images)
        return super().encode(src_inputs, src_mask, src_langs)

    def forward(self, src_inputs=None, src_pads=None, tgt_inputs=None, src_langs=None, tgt_langs=None,
                tgt_mask=None, pad_idx: int = 0, tgt_positions=None, batch=None, proposals=None,
                log_softmax: bool = False, encode_only: bool = False, **kwargs):
        if isinstance(batch, list):
            batch = batch[0]
        if batch is None or src_inputs is not None:
            # NB reference forwards src_mask= to a parent that has no such kwarg (swallowed by **kwargs,
            # src/image_model.py:318-320): the parent recomputes pads from ids.
            return ImageMassSeq2Seq.forward(self, src_inputs=src_inputs, src_mask=src_pads, tgt_inputs=tgt_inputs,
                                            src_langs=src_langs, tgt_langs=tgt_langs, log_softmax=log_softmax)
        image_embeddings, _ = self.encode(images=batch["images"])
        if encode_only:
            return image_embeddings
        batch_lang = int(tgt_langs[0])
        tgt_langs_t = tgt_langs.unsqueeze(-1).expand(-1, tgt_inputs.size(-1))
        pos = tgt_positions[:, :-1] if tgt_positions is not None else None
        return self._decode_and_project(image_embeddings, src_pads, tgt_inputs, tgt_mask, tgt_langs_t, batch_lang,
                                        pos, log_softmax)


# --------------------------------------------------------------------------- loss / optimizer
class SmoothedNLLLoss(nn.Module):
    """src/loss.py:4-27 (reduce forced False, loss.py:8): per-row label-smoothed NLL on log-probs -> [N,1]."""

    def __init__(self, weight=None, ignore_index=-100, reduce: bool = False, epsilon=0.1):
        super().__init__()
        self.ignore_index = ignore_index
        self.epsilon = epsilon

    def forward(self, input, target):
        if target.dim() == input.dim() - 1:
            target = target.unsqueeze(-1)
        nll = -input.gather(dim=-1, index=target)
        smooth = -input.sum(dim=-1, keepdim=True)
        if self.ignore_index is not None:
            pad = target.eq(self.ignore_index)
            nll = nll.masked_fill(pad, 0.)
            smooth = smooth.masked_fill(pad, 0.)
        eps_i = self.epsilon / input.size(-1)
        return (1. - self.epsilon) * nll + eps_i * smooth


def inverse_sqrt_lr(num_updates: int, lr: float, warmup_updates: int, warmup_init_lr: float = 1e-7) -> float:
    """src/utils.py:141-146."""
    lr_step = (lr - warmup_init_lr) / warmup_updates
    decay = lr * warmup_updates ** 0.5
    if num_updates < warmup_updates:
        return warmup_init_lr + num_updates * lr_step
    return max(warmup_init_lr, min(lr, decay * (num_updates ** -0.5)))


class AdamInverseSqrtWithWarmup(torch.optim.Adam):
    """src/utils.py:105-156: Adam(betas 0.9/0.98 via build_optimizer :14-16), lr starts at warmup_init_lr and
    is updated AFTER each step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, warmup_updates=4000,
                 warmup_init_lr=1e-7):
        super().__init__(params, lr=warmup_init_lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.warmup_updates, self.warmup_init_lr, self.max_lr = warmup_updates, warmup_init_lr, lr
        for g in self.param_groups:
            g['num_updates'] = 0

    def step(self, closure=None):
        super().step(closure)
        for g in self.param_groups:
            g['num_updates'] += 1
            g['lr'] = inverse_sqrt_lr(g['num_updates'], self.max_lr, self.warmup_updates, self.warmup_init_lr)


def train_step(model, optimizer, criterion, batch, clip: float = 1.0):
    """One MT micro-step == src/train_image_mt.py:239-295 with accum=1: forward(log_softmax) -> smoothed NLL
    mean -> backward -> clip_grad_norm_ -> step -> zero_grad.  Returns (loss, ntokens)."""
    pred = model(src_inputs=batch["src_texts"], tgt_inputs=batch["dst_texts"], src_mask=batch["src_pad_mask"],
                 tgt_mask=batch["dst_pad_mask"], src_langs=batch["src_langs"], tgt_langs=batch["dst_langs"],
                 log_softmax=True)
    targets = batch["dst_texts"][:, 1:].contiguous().view(-1)[batch["dst_pad_mask"][:, 1:].contiguous().view(-1)]
    loss = criterion(pred, targets).mean()
    loss.backward()
    # de-duplicated parameter list (shared modules appear once in .parameters())
    torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    optimizer.step()
    optimizer.zero_grad()
    return float(loss.detach()), int(targets.numel())
