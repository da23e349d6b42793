"""Flat parameter store: every unique parameter of a model lives in ONE fp32 master buffer (plus one fp32
gradient buffer with the same offsets, Adam moments and an optional bf16 shadow copy).  The nn.Parameters that
the reference's callers and checkpoints see (``encoder.encoder.layer.0.attention.self.query.weight`` ...) are
VIEWS into that buffer, so

  * the C runtime addresses parameters/gradients by element offset (include/imt_hip.h: imt_stack_desc),
  * q|k|v projections are contiguous and run as one GEMM,
  * grad-norm / Adam / the bf16 shadow / the RCCL all-reduce are single passes over contiguous memory,
  * ``state_dict()`` / ``load_state_dict()`` keep the reference's key names and shapes.

Layout is HBM-friendly: each tensor starts on a 256-byte boundary; the order is the order in which gradients
become final during backward (output layers, decoder top->bottom, encoder top->bottom, embeddings) so that
all-reduce buckets are contiguous ranges.
"""
import weakref
from typing import List

import torch
import torch.nn as nn

ALIGN = 64  # elements (256 B in fp32, 128 B in bf16)


class FlatParams:
    def __init__(self, root: nn.Module):
        self._root = weakref.ref(root)
        self.flat = None       # fp32 master [total]
        self.grad = None       # fp32 grads  [total]
        self.exp_avg = None
        self.exp_avg_sq = None
        self._shadow = None    # bf16 copy
        self._shadow_version = -1
        self._shadow_stale = True
        self.entries = []      # (param, offset, numel)
        self.index = {}        # id(param) -> offset
        self.total = 0
        self.layout_version = 0
        self.grad_generation = 0  # bumped whenever the gradient buffer is zeroed (voids partial sums taken over it)
        self._update_events = []  # (lo, hi, event) of an optimizer step still running on its side stream

    # ------------------------------------------------------------------ layout
    def _ordered_params(self) -> List[nn.Parameter]:
        root = self._root()
        order = root.flat_param_order() if hasattr(root, "flat_param_order") else []
        seen, out = set(), []
        for p in list(order) + list(root.parameters()):
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        return out

    def valid(self, full: bool = False) -> bool:
        """Do the parameters still live in the flat buffer?  Every call compares the data pointer of EVERY entry with the layout
        (one list comprehension against a cached tuple: ~20 us at C1) -- anything that moves a parameter's storage is caught at
        the next call, wherever it was done (``model.encoder.to(...)``, ``p.data = ...``, ``load_state_dict(assign=True)``, a
        re-assigned ``nn.Parameter`` keeps its old object in ``entries`` and is caught by the count below).  The complete check
        (dtypes, number of parameters of the model: 0.7 ms of host time at C1 -- four times per step it was most of the step's
        host path) runs when something that can add or replace parameters has happened (``mark_dirty``: module / parameter
        assignment on the model, ``.to()`` / ``.cuda()``), on every 16th call as a safety net, and on request."""
        if self.flat is None:
            return False
        self._checks = getattr(self, "_checks", 0) + 1
        base = self.flat.data_ptr()
        if getattr(self, "_ptr_base", None) != base:
            self._ptr_base = base
            self._ptrs = tuple(base + 4 * off for _, off, _ in self.entries)
        if tuple(p.data_ptr() for p, _, _ in self.entries) != self._ptrs:
            return False
        if not (full or getattr(self, "_dirty", True) or self._checks % 16 == 0):
            return True
        for p, off, n in self.entries:
            if p.dtype != torch.float32:
                return False
        root = self._root()
        if root is not None and sum(1 for _ in root.parameters()) != len(self.entries):
            return False
        self._dirty = False
        return True

    def mark_dirty(self):
        self._dirty = True

    def ensure(self):
        if not self.valid():
            self.rebuild()
        return self

    def rebuild(self):
        params = self._ordered_params()
        if not params:
            raise RuntimeError("FlatParams: model has no parameters")
        device = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        grad = torch.zeros(total, dtype=torch.float32, device=device)
        entries, index = [], {}
        with torch.no_grad():
            for p, off in zip(params, offs):
                n = p.numel()
                view = flat[off:off + n].view(p.shape)
                view.copy_(p.data.to(device=device, dtype=torch.float32))
                gview = grad[off:off + n].view(p.shape)
                if p.grad is not None:
                    gview.copy_(p.grad.to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = gview
                p._imt_store = weakref.ref(self)
                entries.append((p, off, n))
                index[id(p)] = off
        # carry optimizer moments across a rebuild when the set of parameters is unchanged
        old_m, old_v, old_index = self.exp_avg, self.exp_avg_sq, dict(self.index)
        self.flat, self.grad, self.entries, self.index, self.total = flat, grad, entries, index, total
        self.exp_avg = self.exp_avg_sq = None
        if old_m is not None and set(old_index) == set(index):
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)
            for p, off, n in entries:
                o = old_index[id(p)]
                self.exp_avg[off:off + n].copy_(old_m[o:o + n])
                self.exp_avg_sq[off:off + n].copy_(old_v[o:o + n])
        self._shadow = None
        self._shadow_stale = True
        self._dirty = False
        self._ptr_base = None
        self.layout_version += 1

    def offset(self, p: nn.Parameter) -> int:
        return self.index[id(p)]

    # ------------------------------------------------------------------ views used by the runtime
    def attach_grad_views(self):
        for p, off, n in self.entries:
            g = p.grad
            if g is None or g.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + n].view(p.shape)

    def zero_grad(self):
        self.wait_updates(0)
        self.grad_generation += 1
        if self.grad is not None:
            self.grad.zero_()

    def mark_master_changed(self):
        self._shadow_stale = True

    def wait_updates(self, lo: int = 0):
        """Make the current stream wait for the part of an overlapped optimizer step (AdamInverseSqrtWithWarmup.step(
        overlap_next_forward=True)) that writes parameters at offsets >= lo.  Events complete in the order the segments
        were issued (highest offsets first), so a consumer of [lo, total) waits for every segment reaching above lo."""
        if self._update_events:
            cur = torch.cuda.current_stream()
            for seg_lo, seg_hi, ev in self._update_events:
                if seg_hi > lo:
                    cur.wait_event(ev)
            if lo == 0:
                self._update_events = []

    def site_events(self, mod):
        """Per-site events of an overlapped optimizer step for stack ``mod`` -- [embeddings, layer 0, layer 1, ...] as raw
        hipEvent_t handles (None = nothing to wait for) -- or None when there is no pending step / the bf16 shadow needs a
        full refresh anyway.  The update segments run on ONE side stream, so waiting for the LAST-issued segment that
        overlaps a site's parameters covers the earlier ones."""
        if not self._update_events or self._shadow_stale:
            return None
        key = ("_imt_site_ranges", self.layout_version)
        cached = mod.__dict__.get("_imt_site_ranges")
        if cached is None or cached[0] != key:
            def rng(params):
                ps = [p for p in params if id(p) in self.index]
                return (min(self.offset(p) for p in ps), max(self.offset(p) + p.numel() for p in ps))
            sites = [rng(list(mod.embeddings.parameters()))] + [rng(list(l.parameters())) for l in mod._stack_layers()]
            cached = (key, sites)
            mod.__dict__["_imt_site_ranges"] = cached
        out = []
        for lo, hi in cached[1]:
            last = None
            for seg_lo, seg_hi, ev in self._update_events:  # issue order
                if seg_lo < hi and seg_hi > lo:
                    last = ev
            out.append(None if last is None else last.cuda_event)
        return out

    def params_for(self, dtype: torch.dtype, lo: int = 0) -> torch.Tensor:
        """Flat parameter buffer in the compute dtype (fp32 master itself, or the bf16 shadow, refreshed when the
        master changed through torch in-place ops or a fused optimizer step without shadow write).  ``lo``: the caller
        only reads parameters at offsets >= lo (see wait_updates); None: the caller orders itself against the pending
        update (site_events)."""
        if lo is not None:
            self.wait_updates(lo)
        if dtype == torch.float32:
            return self.flat
        assert dtype == torch.bfloat16
        v = self._version_stamp()
        if self._shadow is None or self._shadow_stale or v != self._shadow_version:
            from . import hip_ops as O
            if self._shadow is None:
                self._shadow = torch.empty(self.total, dtype=torch.bfloat16, device=self.flat.device)
            # the cast reads the WHOLE master buffer and writes the whole shadow: every segment of an optimizer step
            # still running on its side stream must have landed, not only those above `lo`
            self.wait_updates(0)
            O.cast_f32_to_bf16(self.flat, self._shadow)
            self._shadow_version = v
            self._shadow_stale = False
        return self._shadow

    def shadow_buffer_for_optimizer(self):
        """bf16 shadow the fused Adam kernel should write in the same pass (None if no bf16 consumer yet)."""
        return self._shadow

    def _version_stamp(self) -> int:
        # in-place torch ops on a parameter view (load_state_dict, torch optimizers) bump that parameter's
        # version counter (not the flat buffer's); the sum is a cheap "master changed behind our back" stamp.
        return sum(p._version for p, _, _ in self.entries)

    def note_shadow_written_by_optimizer(self):
        self._shadow_version = self._version_stamp()
        self._shadow_stale = False

    def anchor(self) -> torch.Tensor:
        """A 1-element leaf that requires grad: passing it into the stack autograd.Functions makes autograd
        schedule their backward even though the real parameter gradients are written by the kernels."""
        a = self.__dict__.get("_anchor")
        if a is None or a.device != self.flat.device:
            a = torch.zeros(1, device=self.flat.device, requires_grad=True)
            self.__dict__["_anchor"] = a
        return a

    def moments(self):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.flat)
            self.exp_avg_sq = torch.zeros_like(self.flat)
        return self.exp_avg, self.exp_avg_sq


def store_of(module: nn.Module) -> FlatParams:
    """The flat store of the top-level model a stack module belongs to (created on first use)."""
    owner = getattr(module, "_imt_owner", None)
    root = owner() if owner is not None else None
    if root is None:
        root = module
    st = root.__dict__.get("_imt_flat_store")
    if st is None:
        st = FlatParams(root)
        root.__dict__["_imt_flat_store"] = st
    return st
