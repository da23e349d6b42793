"""Image branch of the path -- drop-in for the on-path parts of the reference's ``src/image_model.py``:
``ImageMassSeq2Seq`` (text branch ``:157-183``) and ``ImageCaptioning`` (``:267-377``).

The CNN trunk (torchvision ResNet / Faster-RCNN, ``:14-124``) is OUT of scope (frozen
feature extractor whose pretrained weights need a network fetch); region features ``[B, 49, C]`` enter at the
``fc`` layer: ``ImageHead`` = dropout -> fc (no bias) -> + location_embedding -> dropout (``:35-41,77-78``).
"""
import torch
import torch.nn as nn

from . import hip_ops as O
from .bert_seq2seq import BertDecoderModel
from .mass_seq2seq import MassSeq2Seq
from .seq2seq import future_mask  # noqa: F401


class _ImageHeadFn(torch.autograd.Function):
    """dropout -> fc (no bias) -> + location_embedding -> dropout (src/image_model.py:35-41,77-78) on the HIP kernels:
    ``imt_add_rows_dropout`` (input dropout, fp32 features -> compute dtype), ``imt_gemm`` (fc), ``imt_add_rows_dropout``
    (+ location rows, output dropout).  Backward: the output dropout's mask is regenerated from its seed, the location
    gradient is a column sum over the batch (``imt_colsum``), the fc gradient one TN GEMM into the flat gradient buffer.
    The region features are frozen inputs: no gradient flows to them."""

    @staticmethod
    def forward(ctx, anchor, x, head, dtype, p, seed):
        from .param_store import store_of
        store = store_of(head).ensure()
        flat = store.params_for(dtype)
        w_p, loc_p = head.fc.weight, head.location_embedding.weight
        d, C = w_p.shape
        R = loc_p.shape[0]
        wo, lo = store.offset(w_p), store.offset(loc_p)
        w = flat[wo:wo + d * C].view(d, C)
        loc = flat[lo:lo + R * d].view(R, d)
        B = x.shape[0]
        if x.shape[1] != R:
            raise ValueError("image head: %d regions given, location_embedding has %d" % (x.shape[1], R))
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        xd = O.add_rows_dropout(x.reshape(B * R, C).contiguous(), None, out_dtype=dtype, dropout_p=p, dropout_seed=seed)
        y = O.gemm(xd, w, O.IMT_NT, splitk_ws=O.splitk_workspace(xd.device))
        out = O.add_rows_dropout(y, loc, dropout_p=p, dropout_seed=seed + 1)
        ctx.store, ctx.wo, ctx.lo, ctx.dims, ctx.p, ctx.seed = store, wo, lo, (B, R, d, C), p, seed
        ctx.save_for_backward(xd)
        return out.view(B, R, d)

    @staticmethod
    def backward(ctx, dout):
        (xd,) = ctx.saved_tensors
        store = ctx.store
        B, R, d, C = ctx.dims
        dy = O.add_rows_dropout(dout.to(xd.dtype).reshape(B * R, d).contiguous(), None, dropout_p=ctx.p, dropout_seed=ctx.seed + 1)
        O.colsum(dy.view(B, R * d), store.grad[ctx.lo:ctx.lo + R * d])              # d(location_embedding) += sum over images
        gw = store.grad[ctx.wo:ctx.wo + d * C].view(d, C)
        sk = max(1, min((B * R) // 256, 512 // max(1, ((d + 127) // 128) * ((C + 127) // 128))))
        O.gemm(dy, xd, O.IMT_TN, out=gw, accumulate=(sk == 1), split_k=sk)           # d(fc.weight) += dy^T x
        store.attach_grad_views()
        return None, None, None, None, None, None


class ImageHead(nn.Module):
    """Stands in for ModifiedResnet's head; ``feat_dim`` = channels of the frozen trunk (2048 for depth >= 3)."""

    def __init__(self, feat_dim: int, embed_dim: int, dropout: float = 0.1, regions: int = 49):
        super().__init__()
        self.dropout = dropout
        self.fc = nn.Linear(in_features=feat_dim, out_features=embed_dim, bias=False)
        self.location_embedding = nn.Embedding(regions, embed_dim)
        self.layer_norm = nn.LayerNorm(embed_dim, eps=1e-12)  # present (unused) in the reference, :103
        self.fcnn = None

    def forward(self, grid_hidden, compute_dtype=torch.float32):
        """grid_hidden: region features [B, regions, feat_dim] (the reference's x8.view().permute(), :35-36)."""
        from .param_store import store_of
        x = grid_hidden.to(self.fc.weight.device)
        p = float(self.dropout) if self.training else 0.0
        fixed = getattr(self, "_imt_dropout_seed", None)
        seed = (int(fixed) if fixed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())) if p > 0 else 0
        anchor = store_of(self).ensure().anchor() if torch.is_grad_enabled() else None
        return _ImageHeadFn.apply(anchor, x, self, compute_dtype, p, seed), None


class ImageMassSeq2Seq(MassSeq2Seq):
    def __init__(self, text_processor, freeze_image: bool = False, resnet_depth: int = 1, lang_dec: bool = False,
                 use_proposals: bool = False, tie_embed: bool = False, enc_layer: int = 6, dec_layer: int = 3,
                 embed_dim: int = 768, intermediate_dim: int = 3072, use_obj: bool = True, *,
                 num_attention_heads: int = 12, image_feat_dim: int = None):
        super(ImageMassSeq2Seq, self).__init__(text_processor=text_processor, tie_embed=tie_embed, lang_dec=lang_dec,
                                               use_proposals=use_proposals, enc_layer=enc_layer, dec_layer=dec_layer,
                                               embed_dim=embed_dim, intermediate_dim=intermediate_dim,
                                               freeze_image=freeze_image, resnet_depth=resnet_depth,
                                               num_attention_heads=num_attention_heads)
        if image_feat_dim is None:
            image_feat_dim = 512 if resnet_depth <= 2 else 2048  # fc.in_features of resnet18/34 vs 50+ (:87-97)
        # the reference builds a pretrained torchvision trunk here unconditionally (:136-138, network fetch);
        # the build keeps only the trainable head -- region features enter at `fc`.
        self.image_model = ImageHead(image_feat_dim, self.config.hidden_size, self.config.hidden_dropout_prob)
        self.multimodal_attention_gate = nn.Parameter(torch.zeros(1, self.config.hidden_size).fill_(0.1),
                                                      requires_grad=True)
        self.image_attention_w = nn.Linear(self.config.hidden_size, 1)
        self.encoder_attention_w = nn.Linear(self.config.hidden_size, 1)

    def encode(self, src_inputs, src_mask, src_langs, images=None):
        encoder_states = super().encode(src_inputs, src_mask, src_langs)
        if images is not None:
            if isinstance(images, list):
                images = images[0]
            image_embeddings = self.image_model(images, self._imt_compute_dtype)
            return encoder_states[0], image_embeddings
        return encoder_states

    @staticmethod
    def _un(x):
        return x[0] if isinstance(x, list) else x

    def forward(self, src_inputs=None, src_pads=None, tgt_inputs=None, src_langs=None, tgt_langs=None, pad_idx: int = 0,
                tgt_positions=None, batch=None, neg_samples=None, neg_mask=None, proposals=None,
                log_softmax: bool = False, **kwargs):
        u = self._un
        batch, src_langs, tgt_langs, src_pads = u(batch), u(src_langs), u(tgt_langs), u(src_pads)
        src_inputs, tgt_positions, tgt_inputs, proposals = u(src_inputs), u(tgt_positions), u(tgt_inputs), u(proposals)
        if batch is None:
            return MassSeq2Seq.forward(self, src_inputs=src_inputs, tgt_inputs=tgt_inputs, src_langs=src_langs,
                                       tgt_langs=tgt_langs, pad_idx=pad_idx, tgt_positions=tgt_positions,
                                       proposals=proposals, log_softmax=log_softmax)
        raise NotImplementedError(
            "ImageMassSeq2Seq image+text branch: the reference passes a tuple as encoder_states and cannot run "
            "(src/image_model.py:153 vs :82, SURVEY a16); not part of the hot path")

    def loss_fused(self, src_inputs=None, src_pads=None, tgt_inputs=None, src_langs=None, tgt_langs=None,
                   pad_idx: int = 0, tgt_positions=None, batch=None, proposals=None, epsilon: float = 0.1, **kwargs):
        u = self._un
        if u(batch) is not None:
            raise NotImplementedError("image+text branch is not on the hot path")
        return MassSeq2Seq.loss_fused(self, u(src_inputs), u(tgt_inputs), u(src_langs), tgt_langs=u(tgt_langs),
                                      pad_idx=pad_idx, tgt_positions=u(tgt_positions), epsilon=epsilon,
                                      proposals=u(proposals))


class ImageCaptioning(ImageMassSeq2Seq):
    def __init__(self, text_processor, freeze_image: bool = False, resnet_depth: int = 1, lang_dec: bool = False,
                 use_proposals: bool = False, tie_embed: bool = False, enc_layer: int = 6, dec_layer: int = 3,
                 embed_dim: int = 768, intermediate_dim: int = 3072, use_obj: bool = True, *,
                 num_attention_heads: int = 12, image_feat_dim: int = None):
        super(ImageCaptioning, self).__init__(text_processor=text_processor, tie_embed=tie_embed, lang_dec=lang_dec,
                                              use_proposals=use_proposals, enc_layer=enc_layer, dec_layer=dec_layer,
                                              embed_dim=embed_dim, intermediate_dim=intermediate_dim,
                                              freeze_image=freeze_image, resnet_depth=resnet_depth,
                                              num_attention_heads=num_attention_heads, image_feat_dim=image_feat_dim)
        if use_obj:
            # object stream (Faster-RCNN features, :286-296): parameters are created for checkpoint compatibility,
            # but no object features exist without the detector, so the stream stays inactive (object_fc is None).
            if not lang_dec:
                self.obj_decoder = BertDecoderModel(self.config)
            else:
                import copy
                dec = BertDecoderModel(self.config)
                self.obj_decoder = nn.ModuleList([copy.deepcopy(dec) for _ in text_processor.languages])
            self.multistream_attention_gate = nn.Parameter(torch.zeros(1, self.config.hidden_size).fill_(0.1),
                                                           requires_grad=True)
            self._link_stacks()

    def encode(self, src_inputs=None, src_mask=None, src_langs=None, images=None):
        if images is not None:
            if isinstance(images, list):
                images = images[0]
            return self.image_model(images, self._imt_compute_dtype)
        return MassSeq2Seq.encode(self, src_inputs, src_mask, src_langs)

    def _caption_rows(self, batch, src_pads, tgt_inputs, tgt_langs, tgt_mask, pad_idx, tgt_positions, proposals):
        u = self._un
        tgt_positions, tgt_inputs, tgt_mask, tgt_langs = u(tgt_positions), u(tgt_inputs), u(tgt_mask), u(tgt_langs)
        device = self.encoder.embeddings.word_embeddings.weight.device
        image_embeddings, object_fc = self.encode(images=batch["images"])
        assert tgt_inputs is not None
        tgt_inputs = tgt_inputs.to(device)
        tgt_mask = tgt_mask.to(device)
        batch_lang = int(tgt_langs[0])
        tgt_langs_t = self._lang_grid(tgt_langs, tgt_inputs.size(-1), device)
        pos = tgt_positions[:, :-1].to(device) if tgt_positions is not None else None
        rows = self._decode(image_embeddings, u(src_pads), tgt_inputs, tgt_mask, tgt_langs_t, batch_lang,
                            position_ids=pos, proposals=proposals, pad_idx=pad_idx)
        return rows, tgt_inputs, tgt_mask, batch_lang

    def forward(self, src_inputs=None, src_pads=None, tgt_inputs=None, src_langs=None, tgt_langs=None, tgt_mask=None,
                pad_idx: int = 0, tgt_positions=None, batch=None, proposals=None, log_softmax: bool = False,
                encode_only: bool = False, **kwargs):
        batch = self._un(batch)
        if batch is None or src_inputs is not None:  # text-based input (:318-320)
            return ImageMassSeq2Seq.forward(self, src_inputs=src_inputs, src_mask=src_pads, tgt_inputs=tgt_inputs,
                                            src_langs=src_langs, tgt_langs=tgt_langs, proposals=proposals,
                                            log_softmax=log_softmax)
        if encode_only:
            return self.encode(images=batch["images"])[0]
        rows, _, _, batch_lang = self._caption_rows(batch, src_pads, tgt_inputs, tgt_langs, tgt_mask, pad_idx,
                                                    tgt_positions, proposals)
        return self._project(rows, batch_lang, log_softmax)

    def loss_fused(self, src_inputs=None, src_pads=None, tgt_inputs=None, src_langs=None, tgt_langs=None, tgt_mask=None,
                   pad_idx: int = 0, tgt_positions=None, batch=None, proposals=None, epsilon: float = 0.1, **kwargs):
        batch = self._un(batch)
        if batch is None or src_inputs is not None:
            return ImageMassSeq2Seq.loss_fused(self, src_inputs=src_inputs, tgt_inputs=tgt_inputs, src_langs=src_langs,
                                               tgt_langs=tgt_langs, pad_idx=pad_idx, epsilon=epsilon,
                                               proposals=proposals)
        rows, tgt_inputs, tgt_mask, batch_lang = self._caption_rows(batch, src_pads, tgt_inputs, tgt_langs, tgt_mask,
                                                                    pad_idx, tgt_positions, proposals)
        return self._loss_from_rows(rows, tgt_inputs, tgt_mask, batch_lang, epsilon)
