"""BERT encoder / cross-attending decoder / vocabulary projection with the reference's module API
(``src/bert_seq2seq.py:6-144``), running on the HIP runtime in ``libimt_hip.so``.

The module TREE and parameter names follow HuggingFace ``modeling_bert`` 2.9.0 exactly (so reference checkpoints
load: SURVEY section 8b) -- ``embeddings.{word,position,token_type}_embeddings``, ``embeddings.LayerNorm``,
``{encoder|decoder}.layer.{i}.attention.self.{query,key,value}``, ``.attention.output.{dense,LayerNorm}``,
``.crossattention.*``, ``.intermediate.dense``, ``.output.{dense,LayerNorm}`` -- but the leaf modules only HOLD
parameters (views into the flat store, param_store.py).  ``forward`` of a whole stack is one autograd.Function
that calls ``imt_stack_forward`` / ``imt_stack_backward`` through the C ABI; there is no CPU fallback.
"""
import copy
import ctypes
import weakref

import torch
import torch.nn as nn

from . import _lib as L
from . import hip_ops as O
from .lm_config import BertConfig  # noqa: F401  (re-exported like the reference's star import)
from .param_store import store_of


# --------------------------------------------------------------------------- parameter-holding module tree
class BertEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def ordered_params(self, with_key_value=True):
        s, o = self.self, self.output
        if not with_key_value:  # cross-attention whose key|value projections are grouped across layers (see Seq2Seq)
            return [s.query.weight, s.query.bias, o.dense.weight, o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias]
        return [s.query.weight, s.key.weight, s.value.weight, s.query.bias, s.key.bias, s.value.bias,
                o.dense.weight, o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias]


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.is_decoder = config.is_decoder
        if self.is_decoder:  # HF 2.9.0: every is_decoder layer owns a crossattention block
            self.crossattention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def ordered_params(self, with_self_attention=True, with_cross_key_value=True):
        ps = []
        if self.is_decoder:
            ps += self.crossattention.ordered_params(with_key_value=with_cross_key_value)
        ps += [self.intermediate.dense.weight, self.intermediate.dense.bias, self.output.dense.weight,
               self.output.dense.bias, self.output.LayerNorm.weight, self.output.LayerNorm.bias]
        if with_self_attention:
            ps += self.attention.ordered_params()
        return ps


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])


def _init_bert_weights(module, std):
    """HF 2.9.0 BertPreTrainedModel._init_weights: N(0, initializer_range) weights, zero biases, LN = (1, 0)."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=std)
    elif isinstance(module, nn.LayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()


class BertOutputLayer(nn.Module):
    """src/bert_seq2seq.py:6-12: Linear(hidden, vocab) on the selected rows."""

    def __init__(self, config):
        super().__init__()
        self.layer = nn.Linear(config.hidden_size, config.vocab_size)

    def forward(self, input):
        lead = input.shape[:-1]
        x2 = input.reshape(-1, input.shape[-1])
        out = _LinearFn.apply(x2, self.layer.weight, self.layer.bias, self)
        return out.view(*lead, out.shape[-1])


# --------------------------------------------------------------------------- runtime glue
def _contig_after(store, first, rest):
    """q|k|v must be contiguous in the flat buffer (they are allocated that way; verify)."""
    off = store.offset(first)
    n = first.numel()
    for p in rest:
        if store.offset(p) != off + n:
            raise L.ImtError("flat layout broken: q|k|v parameters are not contiguous")
        n += p.numel()
    return off


def _attn_block(store, att: BertAttention, split_kv: bool = False):
    """Offsets of one attention block.  split_kv (cross-attention only): key|value need not follow the query projection;
    returns (block, kv_w, kv_b) with kv_* = -1 when they do follow it."""
    s, o = att.self, att.output
    b = L.AttnBlock()
    kv_w = kv_b = -1
    follows = (store.offset(s.key.weight) == store.offset(s.query.weight) + s.query.weight.numel()
               and store.offset(s.key.bias) == store.offset(s.query.bias) + s.query.bias.numel())
    if split_kv and not follows:
        b.qkv_w, b.qkv_b = store.offset(s.query.weight), store.offset(s.query.bias)
        kv_w = _contig_after(store, s.key.weight, [s.value.weight])
        kv_b = _contig_after(store, s.key.bias, [s.value.bias])
    else:
        b.qkv_w = _contig_after(store, s.query.weight, [s.key.weight, s.value.weight])
        b.qkv_b = _contig_after(store, s.query.bias, [s.key.bias, s.value.bias])
    b.o_w, b.o_b = store.offset(o.dense.weight), store.offset(o.dense.bias)
    b.ln_g, b.ln_b = store.offset(o.LayerNorm.weight), store.offset(o.LayerNorm.bias)
    return (b, kv_w, kv_b) if split_kv else b


class _Pretrained(nn.Module):
    """Shared behaviour of the two stacks (stands in for HF BertPreTrainedModel)."""

    def init_weights(self):
        std = self.config.initializer_range
        self.apply(lambda m: _init_bert_weights(m, std))

    @staticmethod
    def _tie_or_clone_weights(output_embeddings, input_embeddings):
        """HF 2.9.0 modeling_utils._tie_or_clone_weights: the FIRST argument receives the second's weight."""
        output_embeddings.weight = input_embeddings.weight
        if hasattr(output_embeddings, "out_features") and hasattr(input_embeddings, "num_embeddings"):
            output_embeddings.out_features = input_embeddings.num_embeddings

    @property
    def device(self):
        return self.embeddings.word_embeddings.weight.device

    @property
    def dtype(self):
        return self.embeddings.word_embeddings.weight.dtype

    # ---- compute dtype (fp32 parity mode / bf16 MFMA mode); set on the owning model, default fp32
    def compute_dtype(self):
        owner = getattr(self, "_imt_owner", None)
        root = owner() if owner is not None else None
        return getattr(root if root is not None else self, "_imt_compute_dtype", torch.float32)

    def _stack_layers(self):
        raise NotImplementedError

    def _desc(self, store, dtype):
        """ctypes imt_stack_desc for the current flat layout (cached per layout version / dtype)."""
        key = (store.layout_version, dtype, id(store))
        cache = self.__dict__.setdefault("_imt_desc_cache", {})
        if cache.get("key") == key:
            return cache["desc"], cache["layers"]
        cfg = self.config
        layers = self._stack_layers()
        arr = (L.LayerDesc * max(1, len(layers)))()
        for i, lyr in enumerate(layers):
            ld = arr[i]
            ld.self_attn = _attn_block(store, lyr.attention)
            ld.cross_kv_w = ld.cross_kv_b = -1
            if getattr(lyr, "is_decoder", False):
                ld.cross_attn, ld.cross_kv_w, ld.cross_kv_b = _attn_block(store, lyr.crossattention, split_kv=True)
            else:
                ld.cross_attn.qkv_w = -1
            ld.ff1_w, ld.ff1_b = store.offset(lyr.intermediate.dense.weight), store.offset(lyr.intermediate.dense.bias)
            ld.ff2_w, ld.ff2_b = store.offset(lyr.output.dense.weight), store.offset(lyr.output.dense.bias)
            ld.ln2_g, ld.ln2_b = store.offset(lyr.output.LayerNorm.weight), store.offset(lyr.output.LayerNorm.bias)
        d = L.StackDesc()
        d.dtype = O.IMT_BF16 if dtype == torch.bfloat16 else O.IMT_F32
        d.d, d.heads, d.ff = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size
        d.vocab, d.max_pos, d.n_types = cfg.vocab_size, cfg.max_position_embeddings, cfg.type_vocab_size
        d.n_layers = len(layers)
        d.is_decoder = int(bool(getattr(cfg, "is_decoder", False)))
        d.pad_id = cfg.pad_token_id if cfg.pad_token_id is not None else -1
        d.ln_eps = cfg.layer_norm_eps
        d.hidden_dropout, d.attn_dropout = cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob
        e = self.embeddings
        d.emb_word, d.emb_pos = store.offset(e.word_embeddings.weight), store.offset(e.position_embeddings.weight)
        d.emb_type = store.offset(e.token_type_embeddings.weight)
        d.emb_ln_g, d.emb_ln_b = store.offset(e.LayerNorm.weight), store.offset(e.LayerNorm.bias)
        d.layers = ctypes.cast(arr, ctypes.POINTER(L.LayerDesc))
        cache.update(key=key, desc=d, layers=arr)
        return d, arr

    def _run(self, ids, type_ids, pos_ids, key_mask, query_mask, mask3d, causal, enc_states, enc_mask):
        store = store_of(self).ensure()
        dtype = self.compute_dtype()
        dev = store.flat.device
        if dev.type != "cuda":
            raise L.ImtError("imagetranslate_amd: model parameters are on %s; the HIP path needs the GPU "
                             "(no CPU fallback) -- call model.cuda() first" % dev)
        if ids.dim() != 2:
            raise ValueError("input_ids must be [batch, length], got %s" % (tuple(ids.shape),))
        # the runtime takes raw pointers: every operand's shape is checked HERE against what the kernels will index
        # (a [batch] token_type_ids, as Seq2Seq.forward holds before it expands it, would otherwise be read out of bounds)
        B_, T_ = ids.shape
        def shaped(t, name, *shape):
            if t is not None and tuple(t.shape) != shape:
                raise ValueError("%s must be %s, got %s" % (name, list(shape), list(t.shape)))
        if type_ids is not None and tuple(type_ids.shape) in ((B_,), (B_, 1)):
            type_ids = type_ids.reshape(B_, 1).expand(B_, T_)   # one language id per sentence (src/seq2seq.py:96)
        if pos_ids is not None and tuple(pos_ids.shape) in ((T_,), (1, T_)):
            pos_ids = pos_ids.reshape(1, T_).expand(B_, T_)      # HF broadcasts a [1, length] position_ids
        shaped(type_ids, "token_type_ids", B_, T_)
        shaped(pos_ids, "position_ids", B_, T_)
        shaped(key_mask, "attention mask", B_, T_)
        shaped(query_mask, "tgt_query_mask", B_, T_)
        shaped(mask3d, "3-D attention mask", B_, T_, T_)
        if enc_states is not None:
            if enc_states.dim() != 3 or enc_states.shape[0] != B_ or enc_states.shape[2] != self.config.hidden_size:
                raise ValueError("encoder_states must be [%d, source_len, %d], got %s" % (B_, self.config.hidden_size, list(enc_states.shape)))
            shaped(enc_mask, "encoder_attention_mask", B_, enc_states.shape[1])
        i64 = lambda t: None if t is None else t.to(device=dev, dtype=torch.long).contiguous()   # the kernels index int64_t
        ids, type_ids, pos_ids = i64(ids), i64(type_ids), i64(pos_ids)
        def u8(m):  # masks as bytes: a bool tensor already is one byte per element (reinterpret, no conversion kernel)
            if m is None:
                return None
            m = m.to(device=dev)
            if m.dtype == torch.bool:
                return m.contiguous().view(torch.uint8)
            return m.to(dtype=torch.uint8).contiguous()
        anchor = store.anchor() if torch.is_grad_enabled() else None
        # dropout seed of this forward (the backward regenerates the same masks from it); tests pin it
        fixed = getattr(self, "_imt_dropout_seed", None)
        seed = (int(fixed) if fixed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())) if self.training else 0
        return _StackFn.apply(anchor, enc_states, self, store, dtype, ids, type_ids, pos_ids, u8(key_mask), u8(query_mask),
                              u8(mask3d), bool(causal), u8(enc_mask), bool(self.training), seed)


class _StackFn(torch.autograd.Function):
    """One encoder or decoder stack: forward = imt_stack_forward, backward = imt_stack_backward.
    `anchor` (the flat master buffer, requires_grad) only makes autograd schedule the backward; parameter gradients
    are accumulated by the kernels straight into the flat gradient buffer the parameters' .grad views alias."""

    @staticmethod
    def forward(ctx, anchor, enc_states, mod, store, dtype, ids, type_ids, pos_ids, key_mask, query_mask, mask3d, causal,
                enc_mask, training, seed):
        desc, _keep = mod._desc(store, dtype)
        B, T = ids.shape
        dev = ids.device
        io = L.StackIO()
        io.B, io.T, io.training = B, T, int(training)
        io.ids = ids.data_ptr()
        io.type_ids = type_ids.data_ptr() if type_ids is not None else None
        io.pos_ids = pos_ids.data_ptr() if pos_ids is not None else None
        io.key_mask = key_mask.data_ptr() if key_mask is not None else None
        io.query_mask = query_mask.data_ptr() if query_mask is not None else None
        io.mask3d = mask3d.data_ptr() if mask3d is not None else None
        io.causal = int(causal)
        Tk = 0
        if desc.is_decoder:
            if enc_states is None:
                raise L.ImtError("decoder needs encoder_states")
            enc_states = enc_states.to(dtype).contiguous()
            Tk = enc_states.shape[1]
            io.enc_states = enc_states.data_ptr()
            io.enc_mask = enc_mask.data_ptr() if enc_mask is not None else None
        io.Tk = Tk
        io.dropout_seed = seed
        # lowest flat offset this stack reads (its own parameters come first, the shared encoder blocks / embeddings
        # later): lets an optimizer step that is still running on its side stream finish the other segments meanwhile
        lo_key = ("_imt_params_lo", store.layout_version)
        if mod.__dict__.get("_imt_params_lo_key") != lo_key:
            mod.__dict__["_imt_params_lo"] = min(store.offset(p) for p in mod.parameters())
            mod.__dict__["_imt_params_lo_key"] = lo_key
        ev_arr = None
        handles = store.site_events(mod) if not desc.is_decoder else None
        if handles is not None:
            # the encoder is the first consumer of an overlapped optimizer step: instead of waiting for its whole parameter
            # range here, hand the runtime one event per site (embeddings, layer 0, 1, ...) -- the update runs ahead of the
            # forward layer by layer (FlatParams.site_events)
            params = store.params_for(dtype, lo=None)
            ev_arr = (ctypes.c_void_p * len(handles))(*handles)
            io.wait_events = ctypes.cast(ev_arr, ctypes.POINTER(ctypes.c_void_p))
            io.n_wait_events = len(handles)
        else:
            params = store.params_for(dtype, lo=mod.__dict__["_imt_params_lo"])
        desc.params = params.data_ptr()
        desc.grads = store.grad.data_ptr()
        lib = L.load()
        ws_bytes = lib.imt_stack_workspace_bytes(ctypes.byref(desc), B, T, Tk)
        if ws_bytes < 0:
            raise L.ImtError("imt_stack_workspace_bytes failed")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        out = torch.empty((B, T, desc.d), dtype=dtype, device=dev)
        io.out = out.data_ptr()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.imt_stack_forward(ctypes.byref(desc), ctypes.byref(io), ctypes.c_void_p(ws.data_ptr()), ws_bytes, st),
                "imt_stack_forward")
        io.wait_events, io.n_wait_events = None, 0  # (forward only; the handles' array dies with this frame)
        ctx.mod, ctx.store, ctx.dtype, ctx.io, ctx.ws, ctx.ws_bytes = mod, store, dtype, io, ws, ws_bytes
        ctx.keep = (ids, type_ids, pos_ids, key_mask, query_mask, mask3d, enc_mask, enc_states, params)
        # the output goes through save_for_backward: held as a plain attribute it would close a reference cycle
        # (out -> grad_fn -> ctx -> out) that only the cyclic collector frees, ~B*T*d bytes per step until then
        ctx.save_for_backward(out)
        ctx.layout_version = store.layout_version
        ctx.is_decoder = bool(desc.is_decoder)
        return out

    @staticmethod
    def backward(ctx, d_out):
        store, mod, dtype, io = ctx.store, ctx.mod, ctx.dtype, ctx.io
        if store.layout_version != ctx.layout_version:
            raise L.ImtError("parameter layout changed between forward and backward")
        desc, _keep = mod._desc(store, dtype)
        desc.params = ctx.keep[-1].data_ptr()
        desc.grads = store.grad.data_ptr()
        (out,) = ctx.saved_tensors
        io.out = out.data_ptr()
        d_out = d_out.to(dtype).contiguous()
        io.d_out = d_out.data_ptr()
        d_enc = None
        if ctx.is_decoder:
            enc_states = ctx.keep[7]
            d_enc = torch.empty_like(enc_states)
            io.d_enc_states = d_enc.data_ptr()
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib = L.load()
        hook = getattr(store, "segment_hook", None)
        n = desc.n_layers
        if hook is None:
            L.check(lib.imt_stack_backward(ctypes.byref(desc), ctypes.byref(io), ctypes.c_void_p(ctx.ws.data_ptr()),
                                           ctx.ws_bytes, 0, n, st), "imt_stack_backward")
        else:
            # one segment per layer so the data-parallel wrapper can launch all-reduce buckets in between
            for l in range(n - 1, -1, -1):
                L.check(lib.imt_stack_backward(ctypes.byref(desc), ctypes.byref(io), ctypes.c_void_p(ctx.ws.data_ptr()),
                                               ctx.ws_bytes, l, l + 1, st), "imt_stack_backward")
                hook(mod, l)
            if n == 0:
                L.check(lib.imt_stack_backward(ctypes.byref(desc), ctypes.byref(io), ctypes.c_void_p(ctx.ws.data_ptr()),
                                               ctx.ws_bytes, 0, 0, st), "imt_stack_backward")
        store.attach_grad_views()
        ctx.ws = ctx.keep = None
        done = getattr(store, "stack_done_hook", None)
        if done is not None:
            done(mod, ctx.is_decoder)  # e.g. the optimizer's partial gradient norm of everything but the encoder's range
        return (None, d_enc) + (None,) * 13


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b for a parameter pair living in the flat store (BertOutputLayer, image fc)."""

    @staticmethod
    def forward(ctx, x, weight, bias, holder):
        store = store_of(holder).ensure()
        if id(weight) not in store.index:
            raise L.ImtError("parameter is not part of its model's flat store")
        dtype = x.dtype
        flat = store.params_for(dtype)
        wo = store.offset(weight)
        N, K = weight.shape
        w = flat[wo:wo + N * K].view(N, K)
        b = None
        if bias is not None:
            bo = store.offset(bias)
            b = flat[bo:bo + N]
        x = x.contiguous()
        y = O.gemm(x, w, O.IMT_NT, bias=b)
        ctx.store, ctx.wo, ctx.bo, ctx.shape = store, wo, (store.offset(bias) if bias is not None else -1), (N, K)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        store = ctx.store
        N, K = ctx.shape
        dy = O.rows16(dy.to(x.dtype))  # [M, out_features]: rows stay 16-byte aligned for any vocabulary size
        M = dy.shape[0]
        dx = O.gemm(dy, w, O.IMT_NN) if ctx.needs_input_grad[0] else None
        gw = store.grad[ctx.wo:ctx.wo + N * K].view(N, K)
        sk = max(1, min(M // 256, 512 // max(1, ((N + 127) // 128) * ((K + 127) // 128))))
        O.gemm(dy, x, O.IMT_TN, out=gw, accumulate=(sk == 1), split_k=sk,
               a_colsum=store.grad[ctx.bo:ctx.bo + N] if ctx.bo >= 0 else None)
        store.attach_grad_views()
        return dx, None, None, None


# --------------------------------------------------------------------------- the two stacks
class BertEncoderModel(_Pretrained):
    """src/bert_seq2seq.py:94-144."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.init_weights()

    def _stack_layers(self):
        return list(self.encoder.layer)

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                inputs_embeds=None):
        if inputs_embeds is not None:
            raise ValueError("inputs_embeds is not supported by the HIP path")
        if input_ids is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        if attention_mask is not None and attention_mask.dim() == 3:
            return self._run(input_ids, token_type_ids, position_ids, None, None, attention_mask, False, None, None)
        return self._run(input_ids, token_type_ids, position_ids, attention_mask, None, None, False, None, None)


class BertDecoderModel(_Pretrained):
    """src/bert_seq2seq.py:15-91."""

    def __init__(self, config):
        super().__init__()
        self.config = copy.deepcopy(config)
        self.config.is_decoder = True
        self.embeddings = BertEmbeddings(self.config)
        self.decoder = BertEncoder(self.config)
        self.init_weights()

    def _stack_layers(self):
        return list(self.decoder.layer)

    def forward(self, input_ids=None, encoder_attention_mask=None, tgt_attention_mask=None, token_type_ids=None,
                position_ids=None, head_mask=None, inputs_embeds=None, encoder_states=None, *, tgt_query_mask=None):
        """``tgt_attention_mask``: None / 2-D [B,T] (-> causal AND key mask, HF is_decoder semantics) / 3-D
        [B,T,T] (used as is).  ``tgt_query_mask`` (build extension): the ``tgt_mask`` factor of
        ``future_mask`` -- equivalent to passing ``future_mask(tgt_mask)`` as a 3-D mask without materialising it."""
        if inputs_embeds is not None:
            raise ValueError("inputs_embeds is not supported by the HIP path")
        if input_ids is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        if encoder_attention_mask is not None and encoder_attention_mask.dim() != 2:
            raise ValueError("encoder_attention_mask must be [batch, source_len]")
        key_mask = mask3d = None
        causal = False
        if tgt_query_mask is not None:
            causal = True
        elif tgt_attention_mask is None:
            causal = True
        elif tgt_attention_mask.dim() == 2:
            causal, key_mask = True, tgt_attention_mask
        else:
            mask3d = tgt_attention_mask
        return self._run(input_ids, token_type_ids, position_ids, key_mask, tgt_query_mask, mask3d, causal, encoder_states,
                         encoder_attention_mask)
