"""Data-parallel gradient exchange: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI).

Replaces both data-parallel strategies of the reference: torch ``DistributedDataParallel`` at
``src/train_image_mt.py:72-76`` and the threaded ``DataParallelModel`` of ``src/parallel.py`` (per-step parameter
broadcast, GIL-bound -- not reproduced).  Semantics are DDP's: every rank computes the mean loss over its own
tokens, gradients are AVERAGED over ranks, then every rank applies the identical clip + Adam update.

Design for xGMI: the gradients already live in ONE flat fp32 buffer whose layout is the order in which they become
final during backward (param_store.py), so a bucket is a contiguous slice -- no flatten/copy kernels.  The stack
backward is issued layer by layer; after each layer a hook advances the "final prefix" of the flat buffer, and every
bucket that lies inside the prefix is handed to an asynchronous all-reduce (RCCL runs it on its own stream, overlapped
with the remaining backward kernels).  The 1/world_size is folded into the fused clip+Adam kernel (``grad_scale``), so
there is no separate scaling pass.

The bucket boundaries are STATIC: a fixed list of (start, end) slices derived from the flat layout once per layout
version.  Ranks may reach their milestones in different orders (``lang_dec=True``: rank A back-propagates through
decoder[0] while rank B uses decoder[1]), but every rank issues the same sequence of collectives with the same sizes.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .param_store import FlatParams, store_of


class _StreamWork:
    """Handle of a collective enqueued on the communicator's side stream: wait() makes the CURRENT stream wait for it
    (no host blocking), like the Work objects of torch's nccl backend."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class RcclComm:
    """RCCL through the library's own C ABI (imt_comm_*, include/imt_hip.h) -- the exchange without torch.distributed's
    nccl backend in the data path.  The 128-byte unique id travels through whatever torch.distributed group exists (gloo
    or nccl; object broadcast) or, for a single rank, nowhere.  Collectives run on a side HIP stream of the communicator's
    own: launch order on it = call order, the caller's stream is joined by events."""

    def __init__(self, rank: int, world_size: int, group=None):
        import ctypes
        from . import _lib as L
        self._L, self._ct = L, ctypes
        lib = L.load()
        n = lib.imt_comm_unique_id_bytes()
        raw = ctypes.create_string_buffer(n)
        if rank == 0:
            L.check(lib.imt_comm_get_unique_id(raw), "imt_comm_get_unique_id")
        if world_size > 1:
            box = [bytes(raw.raw) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            raw = ctypes.create_string_buffer(box[0], n)
        self.rank, self.world_size = rank, world_size
        self.handle = ctypes.c_void_p()
        L.check(lib.imt_comm_init(raw, world_size, rank, ctypes.byref(self.handle)), "imt_comm_init")
        self.stream = torch.cuda.Stream()

    def _enqueue(self, fn):
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())   # the buffer's producers are enqueued before this point
        self.stream.wait_event(ready)
        with torch.cuda.stream(self.stream):
            fn(self._ct.c_void_p(self.stream.cuda_stream))
            done = torch.cuda.Event()
            done.record(self.stream)
        return _StreamWork(done)

    def _dtype(self, t):
        from . import hip_ops as O
        return O.dt(t)

    def all_reduce_async(self, t: torch.Tensor):
        """In-place sum over the ranks of a contiguous fp32 / bf16 tensor; returns a handle with wait()."""
        assert t.is_cuda and t.is_contiguous()
        lib, L = self._L.load(), self._L
        return self._enqueue(lambda st: L.check(lib.imt_comm_allreduce(self.handle, self._ct.c_void_p(t.data_ptr()), t.numel(),
                                                                       self._dtype(t), st), "imt_comm_allreduce"))

    def broadcast(self, t: torch.Tensor, root: int = 0):
        assert t.is_cuda and t.is_contiguous()
        lib, L = self._L.load(), self._L
        self._enqueue(lambda st: L.check(lib.imt_comm_broadcast(self.handle, self._ct.c_void_p(t.data_ptr()), t.numel(), self._dtype(t),
                                                                root, st), "imt_comm_broadcast")).wait()

    def destroy(self):
        if self.handle:
            torch.cuda.synchronize()
            self._L.check(self._L.load().imt_comm_destroy(self.handle), "imt_comm_destroy")
            self.handle = self._ct.c_void_p()


class GradSync:
    def __init__(self, model, process_group=None, bucket_bytes: int = 32 << 20, broadcast_params: bool = True,
                 comm: Optional[RcclComm] = None):
        """``comm``: an RcclComm -> the buckets go through imt_comm_allreduce (RCCL behind the library's C ABI) instead of
        ``torch.distributed.all_reduce``; also selected by IMT_COMM=rccl when torch.distributed runs the nccl backend."""
        self.model = model
        self.group = process_group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if comm is None and self.world_size > 1 and os.environ.get("IMT_COMM") == "rccl":
            comm = RcclComm(dist.get_rank(process_group), self.world_size, process_group)
        self.comm = comm
        if comm is not None:
            self.world_size = comm.world_size
        if self.world_size > 1 and next(model.parameters()).is_cuda:
            # a collective's resident kernels hold CUs for milliseconds: one-tile-per-CU GEMMs then need a second round on the
            # persistent kernel; the three-workgroups-per-CU kernel degrades gracefully
            from . import _lib as L
            L.load().imt_set_gemm_share_cus(1)
        self._works: List = []
        self._ready = 0       # elements of the flat gradient buffer that are final (prefix)
        self._next = 0        # index of the next bucket of the schedule to launch
        self._schedule = []   # [(start, end)] of this step
        self._schedules = {}  # active head (or None) -> [(start, end)]
        self.store: FlatParams = store_of(model.encoder).ensure()
        self._milestones = self._layer_milestones()
        self._layout_version = self.store.layout_version
        self.store.segment_hook = self._on_segment
        self.store.output_hook = self.output_layers_done
        if broadcast_params and self.world_size > 1:  # DDP ctor broadcast
            if self.comm is not None:
                self.comm.broadcast(self.store.flat, 0)
            else:
                dist.broadcast(self.store.flat, src=0, group=process_group)
            self.store.mark_master_changed()
        self.launched_buckets = []  # (start, end) of the last step, for tests / tuning

    # ------------------------------------------------------------------ layout -> readiness milestones
    def _output_layers(self):
        m = self.model
        return list(m.output_layer) if isinstance(m.output_layer, torch.nn.ModuleList) else [m.output_layer]

    def _layer_milestones(self):
        """(stack module id, layer index) -> end offset of the flat prefix that is final once that layer's
        backward segment has been issued.  Relies on flat_param_order(): outputs, decoder layers top->bottom (+ its
        embedding LN), encoder layers top->bottom, embeddings."""
        st, m = self.store, self.model
        ms = {}
        decs = list(m.decoder) if isinstance(m.decoder, torch.nn.ModuleList) else [m.decoder]
        shared = {id(l.attention) for l in m.encoder.encoder.layer}

        def end_of(params):
            return max(st.offset(p) + p.numel() for p in params)

        for dec in decs:
            layers = list(dec.decoder.layer)
            for li, lyr in enumerate(layers):
                ps = lyr.ordered_params(with_self_attention=id(lyr.attention) not in shared, with_cross_key_value=False)
                ms[(id(dec), li)] = end_of(ps)
            # the grouped cross-attention key|value projections get their gradients in the layer-0 segment
            kv = [p for lyr in layers for p in (lyr.crossattention.self.key.weight, lyr.crossattention.self.value.weight,
                                                 lyr.crossattention.self.key.bias, lyr.crossattention.self.value.bias)]
            ms[(id(dec), 0)] = max(ms.get((id(dec), 0), 0), end_of(kv))
            # embedding LN of the decoder becomes final with layer 0 (embedding backward runs in that segment)
            ms[(id(dec), 0)] = max(ms.get((id(dec), 0), 0),
                                   end_of([dec.embeddings.LayerNorm.weight, dec.embeddings.LayerNorm.bias]))
        enc_layers = list(m.encoder.encoder.layer)
        for li, lyr in enumerate(enc_layers):
            ms[(id(m.encoder), li)] = end_of(lyr.ordered_params())
        ms[(id(m.encoder), 0)] = st.total  # embeddings (and anything after them) final at the very end
        return ms

    def bucket_schedule(self, active_head: Optional[int] = None):
        """The static list of (start, end) all-reduce slices of one step.  ``active_head``: index of the ONE vocabulary
        projection every rank uses in this step (batches of one language direction on all ranks) -- the other heads have
        no gradient anywhere and are left out of the exchange, which is what DDP's ``find_unused_parameters=True`` does
        at src/train_image_mt.py:73 for parameters unused on every rank (61.6 MB of fp32 zeros at C1).  None: exchange
        everything (always correct).  Constraint: only for steps that are their own accumulation window (``--acc 1``);
        ``train_step`` refuses the hint inside a window, where a head idle in this micro-step may hold gradients of an
        earlier one."""
        sched = self._schedules.get(active_head)
        if sched is not None:
            return sched
        st = self.store
        ranges = [(0, st.total)]
        outs = self._output_layers()
        if active_head is not None and len(outs) > 1:
            spans = []
            for o in outs:
                lo = min(st.offset(o.layer.weight), st.offset(o.layer.bias))
                hi = max(st.offset(o.layer.weight) + o.layer.weight.numel(), st.offset(o.layer.bias) + o.layer.bias.numel())
                spans.append((lo, hi))
            heads_lo, heads_hi = min(s[0] for s in spans), max(s[1] for s in spans)
            # the heads must be one contiguous run at the front of the buffer with nothing else in between
            covered = sum(hi - lo for lo, hi in spans)
            if heads_lo == 0 and covered + 64 * len(spans) >= heads_hi:
                lo, hi = spans[active_head]
                ranges = [(lo, hi), (heads_hi, st.total)]
        sched = []
        for lo, hi in ranges:
            s = lo
            while s < hi:
                e = min(hi, s + self.bucket_elems)
                if hi - e < self.bucket_elems // 2:  # no tiny tail bucket
                    e = hi
                sched.append((s, e))
                s = e
        self._schedules[active_head] = sched
        return sched

    def exchanged_bytes(self, active_head: Optional[int] = None) -> int:
        return 4 * sum(e - s for s, e in self.bucket_schedule(active_head))

    # ------------------------------------------------------------------ hooks
    def begin_step(self, active_head: Optional[int] = None):
        self._works, self._ready, self._next, self.launched_buckets = [], 0, 0, []
        self.store.ensure()
        if self.store.layout_version != self._layout_version:
            self._milestones = self._layer_milestones()
            self._schedules = {}
            self._layout_version = self.store.layout_version
        self._schedule = self.bucket_schedule(active_head)

    def output_layers_done(self):
        """Call after the loss backward has produced the vocabulary-projection gradients (they sit at the front of the
        flat buffer)."""
        st = self.store
        self._advance(max(st.offset(o.layer.bias) + o.layer.bias.numel() for o in self._output_layers()))

    def _on_segment(self, stack_module, layer_index):
        end = self._milestones.get((id(stack_module), layer_index))
        if end is not None:
            self._advance(end)

    def _advance(self, ready_end: int):
        if self.world_size <= 1:
            return
        self._ready = max(self._ready, ready_end)
        while self._next < len(self._schedule) and self._schedule[self._next][1] <= self._ready:
            s, e = self._schedule[self._next]
            if self.comm is not None:
                self._works.append(self.comm.all_reduce_async(self.store.grad[s:e]))
            else:
                self._works.append(dist.all_reduce(self.store.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.launched_buckets.append((s, e))
            self._next += 1

    def finish(self) -> float:
        """Launch whatever is left of the schedule, wait for every all-reduce; returns the grad_scale (1/world_size) to
        hand to the fused optimizer step."""
        if self.world_size > 1:
            self._advance(self.store.total)
            for w in self._works:
                w.wait()
            self._works = []
        return 1.0 / self.world_size


def _loss_of(model, batch, epsilon):
    from .mass_seq2seq import MassSeq2Seq
    if not isinstance(model, MassSeq2Seq):  # plain Seq2Seq takes explicit masks (src/seq2seq.py:146)
        return model.loss_fused(batch["src_texts"], batch["dst_texts"], batch["src_pad_mask"], batch["dst_pad_mask"],
                                batch["src_langs"], batch["dst_langs"], epsilon=epsilon, ntokens=batch.get("ntokens"))
    # the trainer's model class derives the masks from the ids (src/mass_seq2seq.py:24-25)
    return model.loss_fused(src_inputs=batch["src_texts"], tgt_inputs=batch["dst_texts"], src_langs=batch["src_langs"],
                            tgt_langs=batch["dst_langs"], epsilon=epsilon)


def clip_in_place(optimizer, store, max_norm: float, grad_scale: float = 1.0):
    """``clip_grad_norm_`` on the flat gradient buffer, in place (the micro-steps of an accumulation window that do not
    end in an optimizer step: src/train_image_mt.py:291 clips after EVERY backward)."""
    from . import hip_ops as O
    O.clip_scale(store.grad, optimizer._grad_norm_sq(store), float(max_norm), float(grad_scale))


def train_step(model, optimizer, batch, sync: Optional[GradSync] = None, clip: float = 1.0, epsilon: float = 0.1,
               update: bool = True, active_head: Optional[int] = None, loss_weight: float = 1.0):
    """One micro-step of the MT hot path == body of ImageMTTrainer.train_epoch (src/train_image_mt.py:239-295):
    forward -> label-smoothed NLL mean -> backward [-> overlapped all-reduce] -> clip -> (Adam -> zero).

    ``update=False`` is a micro-step inside a gradient-accumulation window (``--acc``): the reference clips the
    ACCUMULATED gradient after every backward (:291) and steps every ``accum`` micro-steps (:292-295); the data-parallel
    exchange also happens per micro-step, as under DDP.  ``loss_weight``: ``mtl_weight`` of the captioning trainer
    (src/train_captioning.py:83).  ``active_head``: see GradSync.bucket_schedule."""
    if sync is not None:
        # an idle head is left out of the exchange but the clip / optimizer kernels scale the WHOLE gradient buffer by 1/world:
        # inside an accumulation window a head that was active in an earlier micro-step would be shrunk again in every
        # micro-step it sits out.  The hint is therefore only valid for steps that are their own window (acc = 1).
        if active_head is not None and sync.world_size > 1 and (not update or getattr(sync, "_window_open", False)):
            raise ValueError("train_step: active_head cannot be combined with gradient accumulation under data parallelism "
                             "(pass active_head=None: exchange every head)")
        sync._window_open = not update
        sync.begin_step(active_head)
    loss, ntokens = _loss_of(model, batch, epsilon)
    (loss if loss_weight == 1.0 else loss * loss_weight).backward()
    scale = 1.0
    if sync is not None:
        scale = sync.finish()
    if update:
        optimizer.step(max_grad_norm=clip, grad_scale=scale, zero_grad=True,
                       overlap_next_forward=os.environ.get("IMT_ADAM_OVERLAP", "0") != "0")
    else:
        clip_in_place(optimizer, store_of(model.encoder).ensure(), clip, scale)
    return loss, ntokens
