"""Trainer-side pieces of the path -- drop-in for the parts of the reference's ``src/utils.py`` that sit inside the
timed train step: ``build_optimizer`` (:14-16), ``AdamInverseSqrtWithWarmup`` (:105-156), gradient clipping
(``train_image_mt.py:291``), plus ``mass_mask`` / ``mass_unmask`` (:41-82, host-side batch construction, kept on the
host exactly as the reference) and ``backward`` (:85-90).

The optimizer works on the model's FLAT buffers: global grad-norm = one reduction kernel, clip + Adam + bf16 shadow
write + grad zeroing = one elementwise kernel (``imt_sumsq`` / ``imt_clip_adam``).
"""
import math
import random
from typing import Dict

import torch
from torch.nn.utils.rnn import pad_sequence

from . import hip_ops as O


def build_optimizer(model, learning_rate, warump_steps):
    return AdamInverseSqrtWithWarmup(model.parameters(), lr=learning_rate, betas=(0.9, 0.98),
                                     warmup_updates=warump_steps)


def backward(loss, optimizer=None, fp16: bool = False):
    # bf16 keeps fp32's exponent range: no loss scaling (the reference needs apex amp.scale_loss for fp16, :85-90)
    loss.backward()


class AdamInverseSqrtWithWarmup(torch.optim.Optimizer):
    """Adam whose lr follows linear warm-up then inverse-sqrt decay, updated AFTER each step (src/utils.py:105-156).

    ``step(max_grad_norm=..., grad_scale=...)`` additionally fuses ``clip_grad_norm_`` (and the 1/world_size of
    data-parallel averaging) into the same kernel; plain ``step()`` after an external
    ``torch.nn.utils.clip_grad_norm_`` (the reference's call sequence) gives the same result.
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, warmup_updates=4000,
                 warmup_init_lr=1e-7):
        if weight_decay != 0:
            raise ValueError("weight_decay is not used by the reference (src/utils.py:14-16) and not supported")
        defaults = dict(lr=warmup_init_lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.warmup_updates = warmup_updates
        self.warmup_init_lr = warmup_init_lr
        warmup_end_lr = lr
        self.lr_step = (warmup_end_lr - warmup_init_lr) / warmup_updates
        self.decay_factor = warmup_end_lr * warmup_updates ** 0.5
        for param_group in self.param_groups:
            param_group['num_updates'] = 0
        self.max_lr = lr
        self._sumsq = None
        self._partial = None
        import os
        self._partial_ok = os.environ.get("IMT_PARTIAL_NORM", "1") != "0"
        self.last_grad_norm_sq = None  # device tensor of the last fused grad-norm^2 (no host sync)

    def get_lr_for_step(self, num_updates):
        if num_updates < self.warmup_updates:
            return self.warmup_init_lr + num_updates * self.lr_step
        return max(self.warmup_init_lr, min(self.max_lr, self.decay_factor * (num_updates ** -0.5)))

    def reset(self):
        for param_group in self.param_groups:
            param_group['num_updates'] = 0

    def _store(self):
        st = None
        n = 0
        for g in self.param_groups:
            for p in g['params']:
                ref = getattr(p, "_imt_store", None)
                s = ref() if ref is not None else None
                if s is None or (st is not None and s is not st):
                    return None
                st = s
                n += 1
        if st is None or n != len(st.entries):
            return None
        return st

    def zero_grad(self, set_to_none: bool = False):
        st = self._store()
        if st is not None and st.valid():
            st.zero_grad()
            st.attach_grad_views()
        else:
            super().zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: float = 0.0, grad_scale: float = 1.0, zero_grad: bool = False,
             overlap_next_forward: bool = False):
        """``overlap_next_forward`` (fused path only): the update runs on a side stream in three segments -- encoder +
        embeddings, decoder, output layers -- and each stack of the NEXT forward waits only for the segments it reads
        (FlatParams.wait_updates), so the HBM-bound update of the decoder / output layers runs under the encoder's
        latency-bound kernels.  Same arithmetic, same order of operations per parameter."""
        if closure is not None:
            raise ValueError("closures are not supported")
        st = self._store()
        if st is None:
            # parameters that do not (all) live in one flat store: generic per-tensor path on the GPU
            self._generic_step(max_grad_norm, grad_scale)
        else:
            st.ensure()
            g = self.param_groups[0]
            step = g['num_updates'] + 1
            m, v = st.moments()
            sumsq = None
            if max_grad_norm and max_grad_norm > 0:
                sumsq = self._grad_norm_sq(st)
                self.last_grad_norm_sq = sumsq
                if self._partial_ok and getattr(st, "stack_done_hook", None) is None:
                    st.stack_done_hook = self._stack_done  # from the next backward on: part of the norm under the backward
            else:
                self._partial = None
            segs = self._segments(st) if overlap_next_forward else None
            if segs is None:
                O.clip_adam(st.flat, st.grad, m, v, st.shadow_buffer_for_optimizer(), sumsq, float(max_grad_norm or 0.0),
                            float(grad_scale), float(g['lr']), g['betas'][0], g['betas'][1], g['eps'], step,
                            zero_grad=zero_grad)
            else:
                main = torch.cuda.current_stream()
                if getattr(self, "_side", None) is None:
                    self._side = torch.cuda.Stream()
                st.wait_updates(0)
                done = torch.cuda.Event()
                done.record(main)           # backward, all-reduce and the grad norm are enqueued before this point
                self._side.wait_event(done)
                sh = st.shadow_buffer_for_optimizer()
                events = []
                with torch.cuda.stream(self._side):
                    for lo, hi in segs:     # highest offsets first: what the next forward needs first
                        O.clip_adam(st.flat[lo:hi], st.grad[lo:hi], m[lo:hi], v[lo:hi], None if sh is None else sh[lo:hi], sumsq,
                                    float(max_grad_norm or 0.0), float(grad_scale), float(g['lr']), g['betas'][0], g['betas'][1],
                                    g['eps'], step, zero_grad=zero_grad)
                        ev = torch.cuda.Event()
                        ev.record(self._side)
                        events.append((lo, hi, ev))
                st._update_events = events
            if zero_grad:
                st.grad_generation += 1  # (a partial gradient norm taken before this point is void)
            if st.shadow_buffer_for_optimizer() is not None:
                st.note_shadow_written_by_optimizer()
            else:
                st.mark_master_changed()
        for param_group in self.param_groups:
            param_group['num_updates'] += 1
            param_group['lr'] = self.get_lr_for_step(param_group['num_updates'])

    @staticmethod
    def _segments(st):
        """[(lo, hi)] of the flat buffer in the order the next forward needs them: embeddings (and whatever follows them),
        encoder layers 0, 1, ... (each its own segment: the encoder forward waits layer by layer, FlatParams.site_events),
        decoder(s), output layers -- None when the model does not expose that structure."""
        root = st._root()
        try:
            enc_layers = list(root.encoder.encoder.layer)
            starts = [min(st.offset(p) for p in lyr.ordered_params()) for lyr in enc_layers]  # flat order: top layer first
            e = root.encoder.embeddings
            emb_lo = min(st.offset(p) for p in e.parameters())
            decs = list(root.decoder) if isinstance(root.decoder, torch.nn.ModuleList) else [root.decoder]
            dec_lo = min(st.offset(p) for d in decs for p in d.parameters())
            enc_lo = min(starts)
        except Exception:
            return None
        if not (0 < dec_lo < enc_lo < emb_lo < st.total) or sorted(starts, reverse=True) != starts:
            return None
        segs = [(emb_lo, st.total)]
        for l in range(len(enc_layers)):  # layer l spans [its first parameter, the first parameter of layer l - 1 / the embeddings)
            segs.append((starts[l], starts[l - 1] if l > 0 else emb_lo))
        segs += [(dec_lo, enc_lo), (0, dec_lo)]
        return segs

    # ------------------------------------------------------------------ gradient norm, part of it under the backward
    def _stack_done(self, mod, is_decoder):
        """FlatParams.stack_done_hook: when the DECODER's backward has been enqueued, every gradient below the encoder's range
        (output layers, decoder) is final -- its share of the squared norm is summed on the side stream while the encoder's
        backward runs; step() then only adds the encoder + embedding range.  Fixed order of the two partial sums:
        deterministic.  Off under a data-parallel exchange (the norm is of the REDUCED gradients) and with IMT_PARTIAL_NORM=0."""
        if not is_decoder or not self._partial_ok:
            return  # (the encoder's backward comes after the decoder's and leaves the partial sum valid)
        self._partial = None
        st = self._store()
        if st is None or getattr(st, "segment_hook", None) is not None:
            return
        segs = self._segments(st)
        if segs is None:
            return
        enc_lo = segs[-2][1]
        self._ensure_norm_buffers(st)
        main = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        ready = torch.cuda.Event()
        ready.record(main)
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            self._sumsq.zero_()
            O.sumsq(st.grad[:enc_lo], self._sumsq, self._sumsq_ws_side)
            done = torch.cuda.Event()
            done.record(self._side)
        self._partial = (enc_lo, done, (st.layout_version, st.grad_generation))

    def _ensure_norm_buffers(self, st):
        if self._sumsq is None or self._sumsq.device != st.flat.device:
            self._sumsq = torch.zeros(1, device=st.flat.device)
            self._sumsq_ws = torch.empty(1024, device=st.flat.device, dtype=torch.float32)
            self._sumsq_ws_side = torch.empty(1024, device=st.flat.device, dtype=torch.float32)

    def _grad_norm_sq(self, st):
        """sum(g^2) over the flat gradient buffer as a device scalar (no host sync)."""
        self._ensure_norm_buffers(st)
        part, self._partial = getattr(self, "_partial", None), None
        if part is not None and part[2] == (st.layout_version, st.grad_generation):
            enc_lo, done, _ = part
            torch.cuda.current_stream().wait_event(done)
            O.sumsq(st.grad[enc_lo:], self._sumsq, self._sumsq_ws)  # adds to the partial sum
        else:
            self._sumsq.zero_()
            O.sumsq(st.grad, self._sumsq, self._sumsq_ws)
        return self._sumsq

    def _generic_step(self, max_grad_norm, grad_scale):
        params = [p for g in self.param_groups for p in g['params'] if p.grad is not None]
        if grad_scale != 1.0:
            for p in params:
                p.grad.mul_(grad_scale)
        if max_grad_norm and max_grad_norm > 0:
            torch.nn.utils.clip_grad_norm_(params, max_grad_norm)
        for g in self.param_groups:
            b1, b2 = g['betas']
            t = g['num_updates'] + 1
            for p in g['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                if not state:
                    state['exp_avg'] = torch.zeros_like(p)
                    state['exp_avg_sq'] = torch.zeros_like(p)
                m, v = state['exp_avg'], state['exp_avg_sq']
                m.mul_(b1).add_(p.grad, alpha=1 - b1)
                v.mul_(b2).addcmul_(p.grad, p.grad, value=1 - b2)
                denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(g['eps'])
                p.addcdiv_(m, denom, value=-g['lr'] / (1 - b1 ** t))


# ----------------------------------------------------------------------------------------- host batch construction
def _mass_span_start(length_hint: int) -> int:
    """Start of the masked span: 20% from position 1, 20% from the middle bound, 60% uniformly in [2, bound]
    (same draw order as the reference, src/utils.py:52-60, so seeded runs produce the same batches)."""
    draw = random.random()
    if draw > 0.8:
        return 1
    if draw > 0.6:
        return length_hint
    return random.randint(2, length_hint) if length_hint >= 2 else 2


def mass_mask(mask_prob, pad_indices, src_text, text_processor) -> Dict:
    """MASS batch construction on the host (semantics of src/utils.py:41-78): per row a contiguous span of
    int(len/2) tokens is hidden from the encoder; the decoder is fed the span shifted right (``to_recover``) with
    its ORIGINAL positions (``positions``); hidden tokens are replaced 80/10/10 by <mask>/random/unchanged."""
    assert 0 < mask_prob < 1
    n_rows, width = src_text.size()
    pad_id = text_processor.pad_token_id()
    span_mask = torch.zeros((n_rows, width), dtype=torch.bool)
    spans, span_pos = [], []
    bounds = pad_indices - (1 - mask_prob) * pad_indices
    for row in range(n_rows):
        span_len = int(pad_indices[row] / 2)
        first = _mass_span_start(int(math.ceil(bounds[row])))
        last = first + span_len
        span_mask[row, first:last] = True
        spans.append(src_text[row, first - 1:last])
        span_pos.append(torch.arange(first - 1, last))
    to_recover = pad_sequence(spans, batch_first=True, padding_value=pad_id)
    positions = pad_sequence(span_pos, batch_first=True, padding_value=int(width) - 1)

    targets = src_text[:, 1:][span_mask[:, 1:]]
    originals = src_text[span_mask]
    n_special = len(text_processor.special_tokens)
    vocab = text_processor.vocab_size()
    mask_id = text_processor.mask_token_id()
    replaced = []
    for k in range(originals.size(0)):
        draw = random.random()
        if draw < 0.8:
            replaced.append(mask_id)
        elif draw < 0.9:
            replaced.append(random.randint(n_special, vocab - 1))
        else:
            replaced.append(int(originals[k]))
    src_text[span_mask] = torch.tensor(replaced, dtype=torch.long)
    return {"src_mask": span_mask, "targets": targets, "src_text": src_text, "to_recover": to_recover,
            "positions": positions, "mask_idx": originals}


def mass_unmask(src_text, src_mask, masked_ids):
    """Undo mass_mask in place after the step (src/utils.py:81-82)."""
    src_text[src_mask] = masked_ids


# ----------------------------------------------------------------------------------------- device batch construction
def mass_mask_device(mask_prob, pad_indices, src_text, text_processor, seed: int) -> Dict:
    """mass_mask on the GPU (imt_mass_mask): same outputs and the same in-place semantics as ``mass_mask`` above, all
    tensors on the device of ``src_text``.  Span starts and the 80/10/10 replacement use the kernel's counter-based
    generator keyed by ``seed`` instead of Python's ``random`` (same procedure, different stream); only the sizes that
    follow from ``pad_indices`` are computed on the host."""
    import ctypes
    from . import _lib as L
    if not src_text.is_cuda:
        raise L.ImtError("mass_mask_device needs src_text on the GPU (use mass_mask on the host)")
    assert 0 < mask_prob < 1
    n_rows, width = src_text.shape
    span_len = (pad_indices.cpu().to(torch.int64) // 2)
    offsets = torch.cumsum(span_len, 0) - span_len
    total, recover_width = int(span_len.sum()), int(span_len.max()) + 1
    dev = src_text.device
    src_text = src_text.contiguous()
    out = {"src_mask": torch.empty((n_rows, width), dtype=torch.uint8, device=dev),
           "to_recover": torch.empty((n_rows, recover_width), dtype=torch.int64, device=dev),
           "positions": torch.empty((n_rows, recover_width), dtype=torch.int64, device=dev),
           "targets": torch.empty(total, dtype=torch.int64, device=dev)}
    pad_dev, off_dev = pad_indices.to(dev, torch.int64).contiguous(), offsets.to(dev)
    a = L.MassArgs()
    a.n_rows, a.width, a.recover_width = n_rows, width, recover_width
    a.n_special, a.vocab = len(text_processor.special_tokens), text_processor.vocab_size()
    a.mask_prob, a.seed = float(mask_prob), int(seed)
    a.mask_id, a.pad_id = text_processor.mask_token_id(), text_processor.pad_token_id()
    a.src_text, a.pad_indices, a.row_offsets = src_text.data_ptr(), pad_dev.data_ptr(), off_dev.data_ptr()
    a.src_mask, a.to_recover = out["src_mask"].data_ptr(), out["to_recover"].data_ptr()
    a.positions, a.targets = out["positions"].data_ptr(), out["targets"].data_ptr()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(L.load().imt_mass_mask(ctypes.byref(a), st), "imt_mass_mask")
    out["src_mask"] = out["src_mask"].bool()
    out["src_text"] = src_text
    out["mask_idx"] = out["targets"]
    out["_row_offsets"] = off_dev
    return out


def mass_unmask_device(masked: Dict):
    """Restore the source ids hidden by mass_mask_device (in place)."""
    import ctypes
    from . import _lib as L
    t, m = masked["src_text"], masked["src_mask"].to(torch.uint8).contiguous()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(L.load().imt_mass_unmask(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(m.data_ptr()),
                                     ctypes.c_void_p(masked["mask_idx"].data_ptr()), ctypes.c_void_p(masked["_row_offsets"].data_ptr()),
                                     t.shape[0], t.shape[1], st), "imt_mass_unmask")
