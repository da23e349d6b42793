#!/usr/bin/env python3
"""imt_beam_step alone on the C1 decoding shape (64 sentences x beam 5, V = 30000): per-kernel device time with the logits
coming from HBM (a rotating set of logit buffers) -- beam_row_topk and beam_merge without their neighbours.  GPU only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagetranslate_amd.hip_ops as O
from imagetranslate_amd import _lib as L
B, beam, V, t_max, step = 64, 5, 30000, 150, 40
rows = B * beam
g = torch.Generator(device="cuda").manual_seed(0)
z = lambda *s, dtype: torch.zeros(*s, dtype=dtype, device="cuda")
logit_sets = [torch.randn(rows, V, device="cuda", generator=g) * 3 for _ in range(8)]
scores, sizes = torch.randn(rows, device="cuda", generator=g), torch.full((rows,), float(step), device="cuda")
eos_in, max_lens = z(rows, dtype=torch.uint8), torch.full((B,), 140, dtype=torch.int64, device="cuda")
hist = torch.randint(6, V, (rows, t_max), device="cuda", generator=g)
slots = torch.arange(rows, dtype=torch.int32, device="cuda").unsqueeze(1).expand(rows, t_max).contiguous()
cs, ci = z(rows, beam, dtype=torch.float32), z(rows, beam, dtype=torch.int32)
o_scores, o_sizes, o_eos = z(rows, dtype=torch.float32), z(rows, dtype=torch.float32), z(rows, dtype=torch.uint8)
o_hist, o_slots = z(rows, t_max, dtype=torch.int64), z(rows, t_max, dtype=torch.int32)
o_parent, o_tok, cnt = z(rows, dtype=torch.int32), z(rows, dtype=torch.int64), z(t_max, dtype=torch.int32)


def call(logits):
    a = L.BeamArgs()
    a.B, a.beam, a.rep, a.V, a.step, a.t_max = B, beam, beam, V, step, t_max
    a.logits, a.ld = logits.data_ptr(), V
    a.scores_in, a.sizes_in, a.eos_in = scores.data_ptr(), sizes.data_ptr(), eos_in.data_ptr()
    a.max_lens, a.hist_in, a.slots_in = max_lens.data_ptr(), hist.data_ptr(), slots.data_ptr()
    a.len_penalty_ratio, a.pad_idx, a.eos = 0.8, 0, 4
    a.cand_scores, a.cand_idx = cs.data_ptr(), ci.data_ptr()
    a.scores_out, a.sizes_out, a.eos_out = o_scores.data_ptr(), o_sizes.data_ptr(), o_eos.data_ptr()
    a.hist_out, a.slots_out, a.parent_out, a.tokens_out = o_hist.data_ptr(), o_slots.data_ptr(), o_parent.data_ptr(), o_tok.data_ptr()
    a.eos_count = cnt.data_ptr()
    O.beam_step(a)


lib = L.load()
for i in range(8):
    call(logit_sets[i])
torch.cuda.synchronize()
lib.imt_prof_enable(1)
for i in range(64):
    call(logit_sets[i % 8])
torch.cuda.synchronize()
rowsb = (L.ProfRow * 16)()
n = lib.imt_prof_report(rowsb, 16)
lib.imt_prof_enable(0)
for r in rowsb[:n]:
    print("%-20s %4d launches  %.1f us each" % (r.kind.decode(), r.launches, 1e3 * r.total_ms / r.launches))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(64):
    call(logit_sets[i % 8])
e1.record()
torch.cuda.synchronize()
print("beam step (two launches) back to back: %.1f us" % (e0.elapsed_time(e1) * 1e3 / 64))
