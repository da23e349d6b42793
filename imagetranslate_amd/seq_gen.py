"""Beam search -- drop-in for the reference's ``src/seq_gen.py`` (``BeamDecoder`` ``:27-242``,
``get_outputs_until_eos`` ``:6-24``), same constructor / ``forward`` signature and outputs.

What differs is HOW a step is computed:
  * the reference re-runs the decoder over the whole prefix at every step (``:164-166``) and re-projects the
    encoder states to cross-attention K/V in every layer of every step.  Here (``kv_cache=True``, the default) one
    step decodes ONE position per hypothesis against a self-attention q|k|v cache and per-sentence cross K/V
    computed once (``imt_decode_begin`` / ``imt_decode_step``).  Beam re-ordering never moves the cache: a slot
    table maps (hypothesis, position) to the cache row of the ancestor that produced it.
  * log-softmax, EOS / length-limit zeroing, length penalty, top-k over beam*V, the PAD overwrites and the
    gather/cat bookkeeping (``:193-227``) are one on-device call (``imt_beam_step``); the host only checks the
    "every hypothesis has EOS" stop condition (``:134-136``) every ``sync_every`` steps.
  * ``kv_cache=False`` keeps the reference's literal per-step recomputation (on the HIP decoder) -- used by the
    tests to show both modes produce the same tokens.
Ties between equal scores (``torch.topk`` leaves them unspecified, and ``:194-196`` creates them on purpose) are
broken towards the lowest flat index.  ``:216`` is taken as floor division (torch 1.4 semantics).
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib as L
from . import hip_ops as O
from .param_store import store_of


def get_outputs_until_eos(eos, outputs, size_limit=None, remove_first_token: bool = False):
    """Rows of ``outputs`` cut before their first ``eos`` (rows without one: cut at ``size_limit[r]``)."""
    if outputs.dim() == 1:
        outputs = outputs.unsqueeze(0)
    outputs = outputs.cpu()
    is_eos = outputs == eos
    has = is_eos.any(dim=1)
    first = is_eos.to(torch.int8).argmax(dim=1)
    begin = 1 if remove_first_token else 0
    cut = []
    for r in range(outputs.size(0)):
        if bool(has[r]):
            end = int(first[r])
        else:
            end = outputs.size(1) if size_limit is None else int(size_limit[r])
        cut.append(outputs[r, begin:end])
    return cut


def _un(x):
    return x[0] if isinstance(x, list) else x


class _BeamState:
    """Device-resident ping-pong state of one search (all sizes fixed up front: r_max = B*beam rows)."""

    def __init__(self, B, beam, t_max, V, device, use_slots):
        r = B * beam
        i64, i32, f32, u8 = torch.int64, torch.int32, torch.float32, torch.uint8
        z = lambda *s, dtype: torch.zeros(*s, dtype=dtype, device=device)
        self.hist = [z(r, t_max, dtype=i64), z(r, t_max, dtype=i64)]
        self.slots = [z(r, t_max, dtype=i32), z(r, t_max, dtype=i32)] if use_slots else [None, None]
        self.scores = [z(r, dtype=f32), z(r, dtype=f32)]
        self.sizes = [z(r, dtype=f32), z(r, dtype=f32)]
        self.eos = [z(r, dtype=u8), z(r, dtype=u8)]
        self.cand_scores = z(r, beam, dtype=f32)
        self.cand_idx = z(r, beam, dtype=i32)
        self.parent = z(r, dtype=i32)
        self.tokens = z(r, dtype=i64)
        self.eos_count = z(t_max, dtype=i32)
        self.cur = 0


class BeamDecoder(nn.Module):
    def __init__(self, seq2seq_model, beam_width: int = 5, max_len_a: float = 1.1, max_len_b: int = 5,
                 len_penalty_ratio: float = 0.8, *, kv_cache: bool = True, sync_every: int = 8):
        super(BeamDecoder, self).__init__()
        self.seq2seq_model = seq2seq_model
        self.beam_width = beam_width
        self.max_len_a = max_len_a
        self.max_len_b = max_len_b
        self.len_penalty_ratio = len_penalty_ratio
        self.kv_cache = kv_cache
        self.sync_every = max(1, int(sync_every))

    def len_penalty(self, lengths: torch.Tensor):
        """GNMT length penalty (https://arxiv.org/abs/1609.08144 section 7); the search itself evaluates it on device."""
        return torch.pow((lengths + 6.0) / 6.0, self.len_penalty_ratio).unsqueeze(-1)

    # ---------------------------------------------------------------- one search
    @torch.no_grad()
    def forward(self, src_inputs=None, src_sizes=None, first_tokens=None, src_mask=None, src_langs=None, tgt_langs=None,
                pad_idx=None, max_len: int = None, unpad_output: bool = True, beam_width: int = None, images=None,
                proposals=None, image_embed=None):
        tgt_langs, first_tokens, src_langs, src_mask = _un(tgt_langs), _un(first_tokens), _un(src_langs), _un(src_mask)
        src_sizes, src_inputs, images, image_embed = _un(src_sizes), _un(src_inputs), _un(images), _un(image_embed)
        proposals = _un(proposals)
        model = self.seq2seq_model.module if hasattr(self.seq2seq_model, "module") else self.seq2seq_model
        if beam_width is None:
            beam_width = self.beam_width
        if pad_idx is None:
            pad_idx = model.text_processor.pad_token_id()
        device = model.encoder.embeddings.word_embeddings.weight.device
        if device.type != "cuda":
            raise L.ImtError("imagetranslate_amd: beam search needs the model on the GPU (no CPU fallback)")
        batch_lang = int(tgt_langs[0])
        if src_inputs is not None:
            batch_size = src_inputs.size(0)
        elif images is not None:
            batch_size = images.size(0)
        elif image_embed is not None:
            batch_size = image_embed.size(0)
        else:
            raise ValueError("BeamDecoder needs src_inputs, images or image_embed")
        if images is not None and max_len is None:
            max_len = 512

        # ---- encoder side (once per search, :94-107)
        enc_mask = None
        if src_inputs is not None and images is None:
            src_mask = src_mask.to(device)
            src_langs_t = src_langs.unsqueeze(-1).expand(-1, src_inputs.size(-1))
            encoder_states = model.encode(src_inputs, src_mask, src_langs_t)[0]
            enc_mask = src_mask.to(torch.uint8).contiguous()
        elif src_inputs is None:
            if image_embed is None:
                encoder_states, obj_feat = model.encode(images=images.to(device))
                if obj_feat is not None:
                    raise NotImplementedError("object-stream decoding needs detector features (SURVEY 8(f) row 4)")
            else:
                encoder_states = image_embed.to(device)
        else:
            raise NotImplementedError(
                "image+text beam search: the reference's multimodal encode cannot run (SURVEY a16)")
        dtype = model._imt_compute_dtype
        encoder_states = encoder_states.to(dtype).contiguous()
        Tk = encoder_states.size(1)

        eos = model.text_processor.sep_token_id()
        V = model.config.vocab_size
        max_pos = model.encoder.embeddings.position_embeddings.num_embeddings
        max_len_func = lambda s: min(int(self.max_len_a * s + self.max_len_b), max_pos)
        if max_len is None:
            max_len = max_len_func(src_inputs.size(1))
        if src_inputs is None:
            max_lens_host = torch.LongTensor([max_len] * batch_size)
        else:
            max_lens_host = torch.LongTensor([max_len_func(int(x)) for x in src_sizes])
        max_lens = max_lens_host.to(device)

        decoder = model.decoder if not model.lang_dec else model.decoder[batch_lang]
        output_layer = model.output_layer if (not model.lang_dec) and model.tie_embed else model.output_layer[batch_lang]

        first_tokens = first_tokens.to(device=device, dtype=torch.int64).contiguous()
        B, beam = batch_size, int(beam_width)
        t_max = max(int(max_len), 2)
        st = _BeamState(B, beam, t_max, V, device, self.kv_cache)
        st.hist[0][:B, 0] = first_tokens
        st.eos[0][:B] = (first_tokens == eos).to(torch.uint8)
        if self.kv_cache:
            st.slots[0][:B, 0] = torch.arange(B, dtype=torch.int32, device=device)
        langs = tgt_langs.to(device=device, dtype=torch.int64)
        type_rows = [langs.contiguous(), torch.repeat_interleave(langs, beam, 0).contiguous()]

        store = store_of(decoder).ensure()
        flat = store.params_for(dtype)
        w_out, b_out = output_layer.layer.weight, output_layer.layer.bias
        wo, bo = store.offset(w_out), store.offset(b_out)
        W = flat[wo:wo + w_out.numel()].view(w_out.shape)
        bias = flat[bo:bo + b_out.numel()]
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        lib = L.load()

        inc = None
        if self.kv_cache:
            inc = _Incremental(lib, decoder, store, dtype, flat, encoder_states, enc_mask, B, beam, t_max, stream)
        logits = O.alloc_rows(B * beam, V, torch.float32, device)  # leading dimension padded to 8 for odd vocabularies
        hidden = torch.empty((B * beam, model.config.hidden_size), dtype=dtype, device=device)

        n_cols = 1
        done_at = None
        if beam == 1 and bool((first_tokens == eos).all()):
            done_at = 0
        for i in range(1, max_len):
            if done_at is not None:
                break
            rep = 1 if i == 1 else beam
            rows = B * rep
            cur, nxt = st.cur, st.cur ^ 1
            if inc is not None:
                ids = first_tokens if i == 1 else st.tokens
                inc.step(i - 1, rows, rep, ids, type_rows[0 if rep == 1 else 1], st.slots[cur], hidden)
                states = hidden[:rows]
            else:
                enc = encoder_states if rep == 1 else torch.repeat_interleave(encoder_states, rep, 0)
                emask = None
                if enc_mask is not None:
                    emask = enc_mask if rep == 1 else torch.repeat_interleave(enc_mask, rep, 0)
                prefix = st.hist[cur][:rows, :i].contiguous()
                types = type_rows[0 if rep == 1 else 1].unsqueeze(-1).expand(-1, i)
                states = decoder(encoder_states=enc, input_ids=prefix, encoder_attention_mask=emask,
                                 tgt_attention_mask=torch.ones_like(prefix), token_type_ids=types)[:, -1, :]
            if model.use_proposals:
                states = model.attend_proposal(states, proposals, pad_idx)
            O.gemm(states.contiguous(), W, O.IMT_NT, bias=bias, out=logits[:rows])

            a = L.BeamArgs()
            a.B, a.beam, a.rep, a.V, a.step, a.t_max = B, beam, rep, V, i, t_max
            a.logits, a.ld = logits.data_ptr(), logits.stride(0)
            a.scores_in, a.sizes_in, a.eos_in = st.scores[cur].data_ptr(), st.sizes[cur].data_ptr(), st.eos[cur].data_ptr()
            a.max_lens, a.hist_in = max_lens.data_ptr(), st.hist[cur].data_ptr()
            a.len_penalty_ratio, a.pad_idx, a.eos = float(self.len_penalty_ratio), int(pad_idx), int(eos)
            a.cand_scores, a.cand_idx = st.cand_scores.data_ptr(), st.cand_idx.data_ptr()
            a.scores_out, a.sizes_out, a.eos_out = st.scores[nxt].data_ptr(), st.sizes[nxt].data_ptr(), st.eos[nxt].data_ptr()
            a.hist_out, a.parent_out, a.tokens_out = st.hist[nxt].data_ptr(), st.parent.data_ptr(), st.tokens.data_ptr()
            if self.kv_cache:
                a.slots_in, a.slots_out = st.slots[cur].data_ptr(), st.slots[nxt].data_ptr()
            a.eos_count = st.eos_count.data_ptr()
            O.beam_step(a)
            st.cur = nxt
            n_cols = i + 1
            # stop condition (:134-136), looked at every `sync_every` steps; columns decoded past it are dropped
            if i % self.sync_every == 0 or i == max_len - 1:
                full = (st.eos_count[1:i + 1] == B * beam).nonzero()
                if full.numel() > 0:
                    done_at = int(full[0, 0]) + 1
        if done_at is not None and done_at >= 1:
            n_cols = min(n_cols, done_at + 1)
        if inc is not None:
            inc.check()   # raises if a one-launch decoder step was abandoned (bounded waits); one sync, the outputs are read next anyway
        outputs = st.hist[st.cur].view(B, beam, t_max)[:, 0, :n_cols]
        if unpad_output:
            return get_outputs_until_eos(eos, outputs, size_limit=max_lens_host)
        outputs = outputs.cpu()
        return [outputs[r] for r in range(outputs.size(0))]


class _Incremental:
    """Owns the caches of one search and issues imt_decode_begin / imt_decode_step."""

    def __init__(self, lib, decoder, store, dtype, flat, encoder_states, enc_mask, B, beam, t_max, stream):
        self.lib, self.stream = lib, stream
        desc, self._keep = decoder._desc(store, dtype)
        desc.params = flat.data_ptr()
        desc.grads = None
        self.desc = desc
        self.flat = flat
        dev = encoder_states.device
        self.B, self.r_max, self.t_max, self.Tk = B, B * beam, t_max, encoder_states.size(1)
        nbytes = lambda n, what: self._positive(n, what)
        self.ws_bytes = nbytes(lib.imt_decode_workspace_bytes(ctypes.byref(desc), self.r_max), "imt_decode_workspace_bytes")
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.cache = torch.empty(nbytes(lib.imt_decode_self_cache_bytes(ctypes.byref(desc), self.r_max, t_max),
                                        "imt_decode_self_cache_bytes"), dtype=torch.uint8, device=dev)
        self.cross = torch.empty(nbytes(lib.imt_decode_cross_bytes(ctypes.byref(desc), B, self.Tk), "imt_decode_cross_bytes"),
                                 dtype=torch.uint8, device=dev)
        self.enc_mask = enc_mask
        self.pos_table = torch.arange(t_max, dtype=torch.int64, device=dev).unsqueeze(1).expand(t_max, self.r_max).contiguous()
        L.check(lib.imt_decode_begin(ctypes.byref(desc), ctypes.c_void_p(encoder_states.data_ptr()), B, self.Tk,
                                     ctypes.c_void_p(self.cross.data_ptr()), stream), "imt_decode_begin")

    @staticmethod
    def _positive(n, what):
        if n <= 0:
            raise L.ImtError("%s failed" % what)
        return n

    def step(self, pos, rows, rep, ids, type_ids, slots, out):
        io = L.DecodeIO()
        io.R, io.rep, io.pos, io.Tk, io.t_max, io.r_max = rows, rep, pos, self.Tk, self.t_max, self.r_max
        io.ids, io.type_ids, io.pos_ids = ids.data_ptr(), type_ids.data_ptr(), self.pos_table[pos].data_ptr()
        io.slots = slots.data_ptr()
        io.enc_mask = self.enc_mask.data_ptr() if self.enc_mask is not None else None
        io.self_cache, io.cross_kv, io.out = self.cache.data_ptr(), self.cross.data_ptr(), out.data_ptr()
        L.check(self.lib.imt_decode_step(ctypes.byref(self.desc), ctypes.byref(io), ctypes.c_void_p(self.ws.data_ptr()),
                                         self.ws_bytes, self.stream), "imt_decode_step")

    def check(self):
        """End of a search: did every one-launch step run to its end (include/imt_hip.h: imt_decode_check)?  Synchronises."""
        L.check(self.lib.imt_decode_check(ctypes.byref(self.desc), self.r_max, ctypes.c_void_p(self.ws.data_ptr()), self.stream),
                "imt_decode_check")
