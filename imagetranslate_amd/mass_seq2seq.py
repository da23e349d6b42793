"""MASS masked-seq2seq forward -- drop-in for the reference's ``src/mass_seq2seq.py:6-60``."""
import torch

from .seq2seq import Seq2Seq, future_mask  # noqa: F401


class MassSeq2Seq(Seq2Seq):
    def _unwrap_mass_args(self, src_inputs, tgt_inputs, src_langs, tgt_positions):
        # threaded-DP convention of the reference: any argument may arrive wrapped in a 1-element list (:14-21)
        if isinstance(tgt_inputs, list):
            assert len(tgt_inputs) == 1
            tgt_inputs = tgt_inputs[0]
            src_langs = src_langs[0]
        if isinstance(src_inputs, list):
            src_inputs = src_inputs[0]
        if isinstance(tgt_positions, list):
            tgt_positions = tgt_positions[0]
        return src_inputs, tgt_inputs, src_langs, tgt_positions

    def _mass_rows(self, src_inputs, tgt_inputs, src_langs, pad_idx, tgt_positions, proposals):
        device = self.encoder.embeddings.word_embeddings.weight.device
        tgt_inputs = tgt_inputs.to(device)
        src_inputs = src_inputs.to(device)
        src_pads = src_inputs != pad_idx
        tgt_mask = tgt_inputs != pad_idx
        src_langs_t = self._lang_grid(src_langs, src_inputs.size(-1), device)
        batch_lang = int(src_langs[0])
        encoder_states = self.encode(src_inputs, src_pads, src_langs_t)[0]
        tgt_langs = self._lang_grid(src_langs, tgt_inputs.size(-1), device)
        pos = tgt_positions[:, :-1].to(device) if tgt_positions is not None else None
        rows = self._decode(encoder_states, src_pads, tgt_inputs, tgt_mask, tgt_langs, batch_lang, position_ids=pos,
                            proposals=proposals, pad_idx=pad_idx)
        return rows, tgt_inputs, tgt_mask, batch_lang

    def forward(self, src_inputs, tgt_inputs, src_langs, tgt_langs=None, pad_idx: int = 0, tgt_positions=None,
                log_softmax: bool = False, proposals=None):
        src_inputs, tgt_inputs, src_langs, tgt_positions = self._unwrap_mass_args(src_inputs, tgt_inputs, src_langs,
                                                                                 tgt_positions)
        if tgt_langs is not None:
            # back-translation / MT loss (:27-30)
            device = self.encoder.embeddings.word_embeddings.weight.device
            tgt_inputs = tgt_inputs.to(device)
            src_pads = src_inputs != pad_idx
            tgt_mask = tgt_inputs != pad_idx
            return Seq2Seq.forward(self, src_inputs=src_inputs, src_mask=src_pads, tgt_inputs=tgt_inputs,
                                   proposals=proposals, tgt_mask=tgt_mask, src_langs=src_langs, tgt_langs=tgt_langs,
                                   log_softmax=log_softmax)
        rows, _, _, batch_lang = self._mass_rows(src_inputs, tgt_inputs, src_langs, pad_idx, tgt_positions, proposals)
        return self._project(rows, batch_lang, log_softmax)

    def loss_fused(self, src_inputs, tgt_inputs, src_langs, tgt_langs=None, pad_idx: int = 0, tgt_positions=None,
                   epsilon: float = 0.1, proposals=None, **kwargs):
        """Fast-path loss for MT (tgt_langs given) or MASS batches; returns (loss, ntokens)."""
        src_inputs, tgt_inputs, src_langs, tgt_positions = self._unwrap_mass_args(src_inputs, tgt_inputs, src_langs,
                                                                                 tgt_positions)
        if tgt_langs is not None:
            device = self.encoder.embeddings.word_embeddings.weight.device
            tgt_inputs = tgt_inputs.to(device)
            return Seq2Seq.loss_fused(self, src_inputs, tgt_inputs, src_inputs != pad_idx, tgt_inputs != pad_idx,
                                      src_langs, tgt_langs, epsilon=epsilon, proposals=proposals)
        rows, tgt_inputs, tgt_mask, batch_lang = self._mass_rows(src_inputs, tgt_inputs, src_langs, pad_idx,
                                                                 tgt_positions, proposals)
        return self._loss_from_rows(rows, tgt_inputs, tgt_mask, batch_lang, epsilon)
