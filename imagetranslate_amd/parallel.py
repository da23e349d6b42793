"""Data-parallel gradient exchange: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI).

Replaces both data-parallel strategies of the reference: torch ``DistributedDataParallel`` at
``src/train_image_mt.py:72-76`` and the threaded ``DataParallelModel`` of ``src/parallel.py`` (per-step parameter
broadcast, GIL-bound -- not reproduced).  Semantics are DDP's: every rank computes the mean loss over its own
tokens, gradients are AVERAGED over ranks, then every rank applies the identical clip + Adam update.

Design for xGMI: the gradients already live in ONE flat fp32 buffer whose layout is the order in which they become
final during backward (param_store.py), so a bucket is a contiguous slice -- no flatten/copy kernels.  The stack
backward is issued layer by layer; after each layer a hook hands the newly final slice to an asynchronous
all-reduce (RCCL runs it on its own stream, overlapped with the remaining backward kernels).  The 1/world_size is
folded into the fused clip+Adam kernel (``grad_scale``), so there is no separate scaling pass.
"""
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from .param_store import FlatParams, store_of


class GradSync:
    def __init__(self, model, process_group=None, bucket_bytes: int = 32 << 20, broadcast_params: bool = True):
        self.model = model
        self.group = process_group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._works: List = []
        self._ready = 0       # elements of the flat gradient buffer that are final (prefix)
        self._launched = 0    # prefix already handed to all-reduce
        self.store: FlatParams = store_of(model.encoder).ensure()
        self._milestones = self._layer_milestones()
        self.store.segment_hook = self._on_segment
        self.store.output_hook = self.output_layers_done
        if broadcast_params and self.world_size > 1:
            dist.broadcast(self.store.flat, src=0, group=process_group)  # DDP ctor broadcast
            self.store.mark_master_changed()
        self.launched_buckets = []  # (start, end) of the last step, for tests / tuning

    # ------------------------------------------------------------------ layout -> readiness milestones
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

    # ------------------------------------------------------------------ hooks
    def begin_step(self):
        self._works, self._ready, self._launched, self.launched_buckets = [], 0, 0, []
        self.store.ensure()
        if self.store.layout_version != getattr(self, "_layout_version", None):
            self._milestones = self._layer_milestones()
            self._layout_version = self.store.layout_version

    def output_layers_done(self):
        """Call after the loss backward has produced the vocabulary-projection gradients (they sit at the front of the
        flat buffer)."""
        m, st = self.model, self.store
        outs = list(m.output_layer) if isinstance(m.output_layer, torch.nn.ModuleList) else [m.output_layer]
        self._advance(max(st.offset(o.layer.bias) + o.layer.bias.numel() for o in outs))

    def _on_segment(self, stack_module, layer_index):
        end = self._milestones.get((id(stack_module), layer_index))
        if end is not None:
            self._advance(end)

    def _advance(self, ready_end: int, flush: bool = False):
        if self.world_size <= 1:
            return
        self._ready = max(self._ready, ready_end)
        while self._ready - self._launched >= self.bucket_elems or (flush and self._ready > self._launched):
            end = self._ready if (flush or self._ready - self._launched < 2 * self.bucket_elems) else \
                self._launched + self.bucket_elems
            view = self.store.grad[self._launched:end]
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.launched_buckets.append((self._launched, end))
            self._launched = end

    def finish(self) -> float:
        """Flush the tail bucket, wait for every all-reduce; returns the grad_scale (1/world_size) to hand to the
        fused optimizer step."""
        if self.world_size > 1:
            self._advance(self.store.total, flush=True)
            for w in self._works:
                w.wait()
            self._works = []
        return 1.0 / self.world_size


def train_step(model, optimizer, batch, sync: Optional[GradSync] = None, clip: float = 1.0, epsilon: float = 0.1):
    """One optimizer step of the MT hot path == body of ImageMTTrainer.train_epoch (src/train_image_mt.py:239-295,
    accum = 1): forward -> label-smoothed NLL mean -> backward [-> overlapped all-reduce] -> clip -> Adam -> zero."""
    if sync is not None:
        sync.begin_step()
    from .mass_seq2seq import MassSeq2Seq
    if not isinstance(model, MassSeq2Seq):  # plain Seq2Seq takes explicit masks (src/seq2seq.py:146)
        loss, ntokens = model.loss_fused(batch["src_texts"], batch["dst_texts"], batch["src_pad_mask"],
                                         batch["dst_pad_mask"], batch["src_langs"], batch["dst_langs"], epsilon=epsilon,
                                         ntokens=batch.get("ntokens"))
    else:  # the trainer's model class derives the masks from the ids (src/mass_seq2seq.py:24-25)
        loss, ntokens = model.loss_fused(src_inputs=batch["src_texts"], tgt_inputs=batch["dst_texts"],
                                         src_langs=batch["src_langs"], tgt_langs=batch["dst_langs"], epsilon=epsilon)
    loss.backward()
    scale = 1.0
    if sync is not None:
        scale = sync.finish()
    optimizer.step(max_grad_norm=clip, grad_scale=scale, zero_grad=True,
                   overlap_next_forward=os.environ.get("IMT_ADAM_OVERLAP", "1") != "0")
    return loss, ntokens
