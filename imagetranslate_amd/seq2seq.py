"""Encoder-decoder MT model -- drop-in for the reference's ``src/seq2seq.py:14-213`` on the HIP runtime.

Same constructor, attribute tree, weight topology (tied embedding tables ``:47-52``, decoder self-attention block
IS the encoder's when depths are equal ``:63-65``, per-language output layers ``:61``), ``encode`` / ``forward``
signatures and checkpoint format.  ``forward`` returns logits / log-probs of the NON-PAD target rows ``[N, V]``
exactly as the reference.  Build extensions (keyword-only, documented deviations):

  * ``num_attention_heads`` (reference hard-codes 12, ``src/lm_config.py:13``),
  * ``set_compute_dtype(torch.bfloat16)``: bf16 storage / fp32-accumulate MFMA mode (default fp32 parity mode),
  * ``loss_fused(...)``: training fast path -- vocabulary projection + log-softmax + label-smoothed NLL
    (``src/loss.py``) + ``.mean()`` without materialising the ``[N, V]`` log-prob matrix in fp32.
"""
import copy
import json
import os
import pickle
import weakref

import torch
import torch.nn as nn

from . import hip_ops as O
from . import lm_config
from .bert_seq2seq import BertConfig, BertDecoderModel, BertEncoderModel, BertOutputLayer, _Pretrained
from .param_store import store_of


def future_mask(tgt_mask):
    """src/seq2seq.py:14-17.  mask[b,i,j] = (j <= i) & tgt_mask[b,i] (masks by QUERY row).  Kept for API
    compatibility; the model itself passes (causal, query_mask) to the attention kernel instead of this tensor."""
    attn_shape = (tgt_mask.size(0), tgt_mask.size(1), tgt_mask.size(1))
    fm = torch.triu(torch.ones(attn_shape, device=tgt_mask.device), diagonal=1).type_as(tgt_mask)
    return ~fm & tgt_mask.unsqueeze(-1)


class _LogSoftmaxFn(torch.autograd.Function):
    """F.log_softmax(dim=-1) on [N,V] logits -> fp32 log-probs (src/seq2seq.py:179-180)."""

    @staticmethod
    def forward(ctx, logits):
        lp, _ = O.log_softmax_fwd(O.rows16(logits))
        ctx.save_for_backward(lp)
        ctx.in_dtype = logits.dtype
        return lp

    @staticmethod
    def backward(ctx, dlp):
        (lp,) = ctx.saved_tensors
        return O.log_softmax_bwd(O.rows16(dlp.float()), lp, ctx.in_dtype)


class _SelectRowsFn(torch.autograd.Function):
    """diag_outputs_flat[tgt_non_mask_flat] (src/seq2seq.py:175-177) with a precomputed index list."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.rows = x.shape[0]
        return O.gather_rows(x, idx)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        # every row selected (no padding): the scatter writes all of dx, no zero fill needed
        alloc = torch.empty if idx.numel() == ctx.rows else torch.zeros
        dx = alloc((ctx.rows, dout.shape[1]), device=dout.device, dtype=dout.dtype)
        O.scatter_rows(dout.contiguous(), idx, dx)
        return dx, None


class _FusedXentFn(torch.autograd.Function):
    """mean_r SmoothedNLL(log_softmax(x W^T + b), t)_r without an fp32 [N,V] matrix: the logits are produced in the
    compute dtype, turned into d(logits) in place by one fused kernel, and consumed by the two backward GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, targets, epsilon, ignore_index):
        store = weight._imt_store()
        store.ensure()
        dtype = x.dtype
        flat = store.params_for(dtype)
        V, K = weight.shape
        wo, bo = store.offset(weight), store.offset(bias)
        w = flat[wo:wo + V * K].view(V, K)
        b = flat[bo:bo + V]
        x = x.contiguous()
        n = x.shape[0]
        logits = O.gemm(x, w, O.IMT_NT, bias=b)
        rows = O.xent_fused_fwd_bwd(logits, targets, epsilon, ignore_index, 1.0 / max(n, 1))  # logits <- dlogits
        ctx.store, ctx.wo, ctx.bo, ctx.shape = store, wo, bo, (V, K)
        ctx.save_for_backward(x, w, logits)
        return O.scaled_sum(rows, 1.0 / max(n, 1))

    @staticmethod
    def backward(ctx, g):
        x, w, dlogits = ctx.saved_tensors
        store = ctx.store
        V, K = ctx.shape
        g = g.float().reshape(1).contiguous()  # upstream scalar stays on the device (no host sync)
        n = dlogits.shape[0]
        tiles256 = ((n + 255) // 256) * ((K + 255) // 256)
        if (V >= 16384 and n >= 2048 and K % 4 == 0 and tiles256 <= 128 and dlogits.dtype == torch.bfloat16
                and os.environ.get("IMT_VOCAB_DX_SPLITK", "1") != "0"):
            # few output tiles, K = vocabulary: K-ranges of 256-tile workgroups into fp32 slabs + one reduce launch
            # (8128 x 512 x 30000: 64 tiles x 4 splits fill the chip at the 256-tile rate)
            splits = max(2, min(8, 256 // tiles256))
            slabs = torch.empty((splits * n, K), device=dlogits.device, dtype=torch.float32)
            dx = O.alloc_rows(n, K, dlogits.dtype, dlogits.device)
            O.gemm(dlogits, w, O.IMT_NN, out=dx, aux=slabs, aux_mode=O.IMT_AUX_SPLITK_WS, split_k=splits, alpha_dev=g)
        else:
            dx = O.gemm(dlogits, w, O.IMT_NN, alpha_dev=g, splitk_ws=O.splitk_workspace(dlogits.device))  # few rows: K ranges (imt_gemm)
        gw = store.grad[ctx.wo:ctx.wo + V * K].view(V, K)
        sk = max(1, min(n // 512, 512 // max(1, ((V + 127) // 128) * ((K + 127) // 128))))
        O.gemm(dlogits, x, O.IMT_TN, out=gw, accumulate=(sk == 1), split_k=sk, alpha_dev=g,
               a_colsum=store.grad[ctx.bo:ctx.bo + V])  # bias gradient fused: dlogits is read once
        store.attach_grad_views()
        hook = getattr(store, "output_hook", None)
        if hook is not None:
            hook()  # data-parallel: the vocabulary-projection gradients are final -> first all-reduce bucket
        return dx, None, None, None, None, None


class Seq2Seq(nn.Module):
    def __init__(self, text_processor, lang_dec: bool = True, use_proposals=False, tie_embed=False,
                 enc_layer: int = 6, dec_layer: int = 3, embed_dim: int = 768, intermediate_dim: int = 3072,
                 freeze_image: bool = False, resnet_depth: int = 1, use_obj: bool = False, *,
                 num_attention_heads: int = 12):
        super(Seq2Seq, self).__init__()
        self.text_processor = text_processor
        self.config = lm_config.get_config(vocab_size=text_processor.tokenizer.get_vocab_size(),
                                           pad_token_id=text_processor.pad_token_id(),
                                           bos_token_id=text_processor.bos_token_id(),
                                           eos_token_id=text_processor.sep_token_id(),
                                           enc_layer=enc_layer, embed_dim=embed_dim, intermediate_dim=intermediate_dim,
                                           num_attention_heads=num_attention_heads)
        self.enc_layer = enc_layer
        self.dec_layer = dec_layer
        self.embed_dim = embed_dim
        self.intermediate_dim = intermediate_dim
        self.num_attention_heads = num_attention_heads
        self.config["type_vocab_size"] = len(text_processor.languages)
        self.config = BertConfig(**self.config)
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
                # NB reference quirk: the tie is applied to the WRAPPER module, which only
                # registers an extra shared parameter `output_layer.weight`; `output_layer.layer.weight` stays untied.
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
        if self.use_proposals:
            self.proposal_embedding = self.encoder.embeddings.word_embeddings
            self.lexical_gate = nn.Parameter(torch.zeros(1, self.config.hidden_size).fill_(0.1), requires_grad=True)
            self.lexical_layer_norm = nn.LayerNorm(self.config.hidden_size, eps=self.config.layer_norm_eps)

        self.freeze_image = freeze_image
        self.resnet_depth = resnet_depth
        self._imt_compute_dtype = torch.float32
        self._link_stacks()

    # ------------------------------------------------------------------ flat-store plumbing
    def _stacks(self):
        out = [self.encoder]
        out += list(self.decoder) if isinstance(self.decoder, nn.ModuleList) else [self.decoder]
        obj = self.__dict__.get("_modules", {}).get("obj_decoder")
        if obj is not None:
            out += list(obj) if isinstance(obj, nn.ModuleList) else [obj]
        return out

    def _link_stacks(self):
        ref = weakref.ref(self)
        holders = list(self._stacks())
        outs = self.__dict__.get("_modules", {}).get("output_layer")
        if outs is not None:
            holders += list(outs) if isinstance(outs, nn.ModuleList) else [outs]
        img = self.__dict__.get("_modules", {}).get("image_model")
        if img is not None:
            holders.append(img)
        for st in holders:
            st.__dict__["_imt_owner"] = ref

    def flat_param_order(self):
        """Order of the flat buffer = order in which gradients become final in backward (bucket-friendly)."""
        ps = []
        outs = list(self.output_layer) if isinstance(self.output_layer, nn.ModuleList) else [self.output_layer]
        for o in outs:
            ps += [o.layer.weight, o.layer.bias]
        decs = list(self.decoder) if isinstance(self.decoder, nn.ModuleList) else [self.decoder]
        shared = {id(l.attention) for l in self.encoder.encoder.layer}
        for dec in decs:
            layers = list(dec.decoder.layer)
            for lyr in reversed(layers):
                ps += lyr.ordered_params(with_self_attention=id(lyr.attention) not in shared, with_cross_key_value=False)
            # the cross-attention key|value projections of ALL layers, contiguous in layer order ([L*2d, d] then [L*2d]):
            # the runtime projects the encoder states for every layer with one GEMM (and one each for d(encoder states)
            # and dW in backward); their gradients are final when the decoder's backward reaches layer 0
            for lyr in layers:
                ps += [lyr.crossattention.self.key.weight, lyr.crossattention.self.value.weight]
            for lyr in layers:
                ps += [lyr.crossattention.self.key.bias, lyr.crossattention.self.value.bias]
            ps += [dec.embeddings.LayerNorm.weight, dec.embeddings.LayerNorm.bias]
        for lyr in reversed(list(self.encoder.encoder.layer)):
            ps += lyr.ordered_params()
        e = self.encoder.embeddings
        ps += [e.LayerNorm.weight, e.LayerNorm.bias, e.position_embeddings.weight, e.token_type_embeddings.weight,
               e.word_embeddings.weight]
        return ps

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._link_stacks()
        st = self.__dict__.get("_imt_flat_store")
        if st is not None:
            st.flat = None  # layout invalid after a device / dtype move; rebuilt lazily
        return out

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        if isinstance(value, (nn.Module, nn.Parameter)):
            st = self.__dict__.get("_imt_flat_store")
            if st is not None:
                st.mark_dirty()  # the set of parameters may have changed: next use re-validates the flat store in full
        if isinstance(value, nn.Module) and "_imt_compute_dtype" in self.__dict__:
            self._link_stacks()  # e.g. caption_model.encoder = mt_model.encoder (train_captioning.py:218-220)

    def set_compute_dtype(self, dtype):
        if dtype in (torch.float16, "fp16", "bf16"):
            dtype = torch.bfloat16  # the reference's --fp16 (apex amp O2) maps to bf16 MFMA on MI355X
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        self._imt_compute_dtype = dtype
        return self

    def zero_grad(self, set_to_none: bool = False):
        st = self.__dict__.get("_imt_flat_store")
        if st is not None and st.flat is not None:
            st.zero_grad()
            st.attach_grad_views()
        else:
            super().zero_grad(set_to_none=set_to_none)

    # ------------------------------------------------------------------ reference API
    def init_from_lm(self, lm):
        raise NotImplementedError("init_from_lm depends on the reference's broken LM class (SURVEY section 2 #17)")

    def encode(self, src_inputs, src_mask, src_langs, images=None):
        device = self.encoder.embeddings.word_embeddings.weight.device
        if src_inputs.device != device:
            src_inputs = src_inputs.to(device)
            src_mask = src_mask.to(device)
            src_langs = src_langs.to(device)
        encoder_states = self.encoder(src_inputs, attention_mask=src_mask, token_type_ids=src_langs)
        return (encoder_states, None)

    def attend_proposal(self, decoder_output, proposals, pad_idx):
        """src/seq2seq.py:110-144 (lexical proposals, off by default): plain torch ops on the GPU, not a kernel
        target.  Includes the reference's no-op mask fill (:132)."""
        device = self.encoder.embeddings.word_embeddings.weight.device
        proposals = proposals.to(device)
        attend_mask = (proposals == pad_idx)
        dt = decoder_output.dtype
        mapped_output = decoder_output.float()
        proposal_embedding = self.proposal_embedding(proposals).float()  # nn.Embedding with padding_idx: the pad row gets no gradient (:116)
        if decoder_output.dim() == 3:
            proposal_embedding = proposal_embedding.unsqueeze(1).expand(-1, decoder_output.size(1), -1, -1)
            mapped_output = mapped_output.unsqueeze(2)
            attend_mask = attend_mask.unsqueeze(1)
        else:
            if len(proposals) < len(decoder_output):
                beam_width = int(len(decoder_output) / len(proposals))
                proposals = torch.repeat_interleave(proposals, beam_width, 0)
                attend_mask = torch.repeat_interleave(attend_mask, beam_width, 0)
                proposal_embedding = torch.repeat_interleave(proposal_embedding, beam_width, 0)
            mapped_output = mapped_output.unsqueeze(1)
        attend_scores = torch.matmul(mapped_output, proposal_embedding.transpose(-1, -2)).squeeze(-2)
        attend_probs = torch.softmax(attend_scores, dim=-1)
        proposal_values = torch.sum(attend_probs.unsqueeze(-1) * proposal_embedding, dim=-2)
        final_proposal_mask = torch.all(proposals == pad_idx, dim=-1)
        proposal_values = proposal_values.masked_fill(final_proposal_mask.view(final_proposal_mask.shape + (1,) * (
            proposal_values.dim() - final_proposal_mask.dim())), 1e-8)
        sig_gate = torch.sigmoid(self.lexical_gate + 1e-8)
        combined = sig_gate * decoder_output.float() + (1 - sig_gate) * proposal_values
        return self.lexical_layer_norm(combined).to(dt)

    # shared tail of every forward variant: decoder -> non-pad row select -> vocabulary projection
    def _uniform_grid(self, n_rows, width, value, device):
        cache = self.__dict__.setdefault("_imt_lang_grids", {})
        key = (int(n_rows), int(width), int(value), str(device))
        grid = cache.get(key)
        if grid is None:
            if len(cache) > 256:
                cache.clear()
                self.__dict__["_imt_grid_info"] = {}
            grid = torch.full((int(n_rows), int(width)), int(value), dtype=torch.int64, device=device)
            cache[key] = grid
            self.__dict__.setdefault("_imt_grid_info", {})[id(grid)] = (int(n_rows), int(value))
        return grid

    def _lang_grid(self, langs, width, device):
        """Per-row language ids [B] -> token-type ids [B, width] on the device (src/seq2seq.py:151-152).  Batches are built
        per language pair, so the ids are almost always uniform: that grid is kept on the device and reused (no expand +
        host->device copy + contiguous copy per step)."""
        if not langs.is_cuda and langs.numel() > 0:
            lo, hi = int(langs.min()), int(langs.max())
            if lo == hi:
                return self._uniform_grid(langs.numel(), width, lo, device)
        return langs.unsqueeze(-1).expand(-1, width).to(device)

    @staticmethod
    def _selection(tgt_inputs, tgt_mask, ntokens=None):
        """(row indices, targets) of the non-pad target positions (src/seq2seq.py:175-177, train_image_mt.py:253-256).
        The row count is data dependent, so this is the step's ONE host synchronisation; the fast path calls it before
        anything is enqueued so that it waits on nothing and the rest of the step is launched without bubbles."""
        if tgt_mask.is_cuda and tgt_inputs.is_cuda and tgt_inputs.dtype == torch.int64 and tgt_mask.dtype in (torch.bool, torch.uint8):
            return O.select_plan(tgt_mask, tgt_inputs, col0=1, count=ntokens)
        idx = torch.nonzero(tgt_mask[:, 1:].reshape(-1), as_tuple=False).view(-1)
        return idx.to(torch.int32), tgt_inputs[:, 1:].reshape(-1)[idx]

    def _decode(self, encoder_states, enc_mask, tgt_inputs, tgt_mask, tgt_langs_t, batch_lang, position_ids=None,
                proposals=None, pad_idx=0, sel_idx=None):
        decoder = self.decoder if not self.lang_dec else self.decoder[batch_lang]
        info = self.__dict__.get("_imt_grid_info", {}).get(id(tgt_langs_t))
        if info is not None:  # a cached uniform grid: use the (T-1)-wide one instead of a non-contiguous slice
            types = self._uniform_grid(info[0], tgt_langs_t.size(1) - 1, info[1], tgt_langs_t.device)
        else:
            types = tgt_langs_t[:, :-1]
        decoder_output = decoder(encoder_states=encoder_states, input_ids=tgt_inputs[:, :-1],
                                 encoder_attention_mask=enc_mask, tgt_query_mask=tgt_mask[:, :-1],
                                 position_ids=position_ids, token_type_ids=types)
        if self.use_proposals:
            decoder_output = self.attend_proposal(decoder_output, proposals, pad_idx)
        flat = decoder_output.reshape(-1, decoder_output.size(-1))
        if sel_idx is None:
            sel_idx = self._selection(tgt_inputs, tgt_mask)[0]
        if sel_idx.numel() == flat.shape[0]:
            return flat  # no padding: the ordered selection of every row is the identity -- no gather / scatter launches
        return _SelectRowsFn.apply(flat, sel_idx)

    def _project(self, rows, batch_lang, log_softmax):
        output_layer = self.output_layer if (not self.lang_dec) and self.tie_embed else self.output_layer[batch_lang]
        outputs = output_layer(rows)
        if log_softmax:
            outputs = _LogSoftmaxFn.apply(outputs)
        return outputs

    def forward(self, src_inputs, tgt_inputs, src_mask, tgt_mask, src_langs, tgt_langs, proposals=None,
                log_softmax: bool = False):
        "Take in and process masked src and target sequences."
        device = self.encoder.embeddings.word_embeddings.weight.device
        batch_lang = int(tgt_langs[0])
        src_langs = src_langs.unsqueeze(-1).expand(-1, src_inputs.size(-1))
        tgt_langs = tgt_langs.unsqueeze(-1).expand(-1, tgt_inputs.size(-1)).to(device)
        src_inputs = src_inputs.to(device)
        src_langs = src_langs.to(device)
        tgt_inputs = tgt_inputs.to(device)
        tgt_mask = tgt_mask.to(device)
        src_mask = src_mask.to(device)
        encoder_states = self.encode(src_inputs, src_mask, src_langs)[0]
        rows = self._decode(encoder_states, src_mask, tgt_inputs, tgt_mask, tgt_langs, batch_lang, proposals=proposals,
                            pad_idx=self.text_processor.pad_token_id())
        return self._project(rows, batch_lang, log_softmax)

    def loss_fused(self, src_inputs, tgt_inputs, src_mask, tgt_mask, src_langs, tgt_langs, epsilon: float = 0.1,
                   proposals=None, ntokens=None):
        """Training fast path == ``SmoothedNLLLoss(ignore_index=pad)(self(..., log_softmax=True), targets).mean()``
        with targets = tgt_inputs[:, 1:][tgt_mask[:, 1:]] (src/train_image_mt.py:249-256,282).
        `ntokens`: int(tgt_mask[:, 1:].sum()) when the caller has it on the host (the reference computes it from the
        batch at train_image_mt.py:253-256; our datasets put it in the batch dict) -- the step then needs no host
        synchronisation at all.  Returns (loss, ntokens)."""
        device = self.encoder.embeddings.word_embeddings.weight.device
        batch_lang = int(tgt_langs[0])
        src_langs_t = self._lang_grid(src_langs, src_inputs.size(-1), device)
        tgt_langs_t = self._lang_grid(tgt_langs, tgt_inputs.size(-1), device)
        src_inputs, tgt_inputs = src_inputs.to(device), tgt_inputs.to(device)
        src_mask, tgt_mask = src_mask.to(device), tgt_mask.to(device)
        sel_idx, targets = self._selection(tgt_inputs, tgt_mask, ntokens)
        encoder_states = self.encode(src_inputs, src_mask, src_langs_t)[0]
        rows = self._decode(encoder_states, src_mask, tgt_inputs, tgt_mask, tgt_langs_t, batch_lang, proposals=proposals,
                            pad_idx=self.text_processor.pad_token_id(), sel_idx=sel_idx)
        return self._loss_from_rows(rows, tgt_inputs, tgt_mask, batch_lang, epsilon, targets=targets)

    def _loss_from_rows(self, rows, tgt_inputs, tgt_mask, batch_lang, epsilon, targets=None):
        if targets is None:
            targets = tgt_inputs[:, 1:][tgt_mask[:, 1:]].contiguous()
        output_layer = self.output_layer if (not self.lang_dec) and self.tie_embed else self.output_layer[batch_lang]
        loss = _FusedXentFn.apply(rows, output_layer.layer.weight, output_layer.layer.bias, targets, float(epsilon),
                                  int(self.text_processor.pad_token_id()))
        return loss, int(targets.numel())

    def state_dict(self, *args, **kwargs):
        st = self.__dict__.get("_imt_flat_store")
        if st is not None:
            st.wait_updates(0)  # an overlapped optimizer step (side stream) must have landed before parameters are read
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        st = self.__dict__.get("_imt_flat_store")
        if st is not None:
            st.wait_updates(0)  # copies into parameter views must not race an optimizer step on its side stream
            st.mark_master_changed()
        return super().load_state_dict(state_dict, *args, **kwargs)

    def save(self, out_dir: str):
        if not os.path.exists(out_dir):
            os.makedirs(out_dir)
        with open(os.path.join(out_dir, "mt_config"), "wb") as fp:
            pickle.dump((self.lang_dec, self.use_proposals, self.enc_layer, self.dec_layer, self.embed_dim,
                         self.intermediate_dim, self.tie_embed, self.resnet_depth, self.freeze_image), fp)
        torch.save({k: v.detach().cpu() for k, v in self.state_dict().items()},
                   os.path.join(out_dir, "mt_model.state_dict"))
        # build extension (the reference hard-codes 12 heads, src/seq2seq.py:37): remembered beside the reference files
        extra = {"num_attention_heads": int(self.config.num_attention_heads)}
        img = self.__dict__.get("_modules", {}).get("image_model")
        if img is not None and hasattr(img, "fc"):  # channels of the region features the `fc` layer was built for
            extra["image_feat_dim"] = int(img.fc.in_features)
        with open(os.path.join(out_dir, "imt_config.json"), "w") as fp:
            json.dump(extra, fp)

    @staticmethod
    def load(cls, out_dir: str, tok_dir: str, use_obj: bool = False, text_processor=None, **kw):
        if text_processor is None:
            from .textprocessor import TextProcessor
            text_processor = TextProcessor(tok_model_path=tok_dir)
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        from .safe_pickle import load_mt_config  # the reference's pickle file, read without executing anything from it
        (lang_dec, use_proposals, enc_layer, dec_layer, embed_dim, intermediate_dim, tie_embed, resnet_depth,
         freeze_image) = load_mt_config(os.path.join(out_dir, "mt_config"))
        extra = os.path.join(out_dir, "imt_config.json")
        if os.path.exists(extra):
            with open(extra, "r") as fp:
                saved = json.load(fp)
            if "num_attention_heads" not in kw and "num_attention_heads" in saved:
                kw["num_attention_heads"] = int(saved["num_attention_heads"])
            import inspect
            if "image_feat_dim" in saved and "image_feat_dim" not in kw and "image_feat_dim" in inspect.signature(cls.__init__).parameters:
                kw["image_feat_dim"] = int(saved["image_feat_dim"])
        mt_model = cls(text_processor=text_processor, lang_dec=lang_dec, use_proposals=use_proposals, tie_embed=tie_embed,
                       enc_layer=enc_layer, dec_layer=dec_layer, embed_dim=embed_dim, intermediate_dim=intermediate_dim,
                       freeze_image=freeze_image, resnet_depth=resnet_depth, use_obj=use_obj, **kw)
        sd = torch.load(os.path.join(out_dir, "mt_model.state_dict"), map_location="cpu", weights_only=True)
        mt_model.load_state_dict(sd, strict=False)
        return mt_model.to(device)
