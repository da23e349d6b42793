"""CPU oracle for MASS batch construction -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

``mass_mask_reference`` restates ``mass_mask`` of the reference (``src/utils.py:41-78``) step by step, with the random
draws injected: ``draw_start(row) -> (u0, u1)`` feeds the span-start rule (``:52-60``: u0 > 0.8 -> 1, u0 > 0.6 -> the
bound, else uniform in [2, bound]) and ``draw_token(row, col) -> (u, r)`` the 80/10/10 replacement (``:70-76``).
``counter_uniform`` is the generator of ``imt_mass_mask`` (csrc/batch.hip) in numpy integer arithmetic.

Pinning: the reference's ``src/utils.py`` cannot be imported here (it imports ``apex`` / ``sacrebleu``-dependent
modules at the top), so the procedure is pinned by its text only -- "parity unpinned"; the test value of this file is
that two independent forms (this loop form, the HIP kernel) agree bit for bit on the same draws.
"""
import math

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

_M = 0xFFFFFFFF


def _mix32(x):
    x &= _M
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M
    x ^= x >> 16
    return x


def counter_uniform(seed: int, stream: int, index: int) -> np.float32:
    h = _mix32((index ^ (seed & _M)) & _M)
    h = _mix32((h + stream * 0x9E3779B9 + (seed >> 32)) & _M)
    return np.float32(h >> 8) * np.float32(1.0 / 16777216.0)


def mass_mask_reference(mask_prob, pad_indices, src_text, n_special, vocab, mask_id, pad_id, draw_start, draw_token):
    src_text = src_text.clone()
    n_rows, width = src_text.shape
    src_mask = torch.zeros((n_rows, width), dtype=torch.bool)
    to_recover, to_recover_pos = [], []
    for r in range(n_rows):
        pad = np.float32(int(pad_indices[r]))
        bound = pad - np.float32(np.float32(1.0) - np.float32(mask_prob)) * pad  # float32 tensor arithmetic
        hint = int(math.ceil(float(bound)))
        span_len = int(int(pad_indices[r]) / 2)
        u0, u1 = draw_start(r)
        if u0 > np.float32(0.8):
            first = 1
        elif u0 > np.float32(0.6):
            first = hint
        elif hint >= 2:
            first = min(hint, 2 + int(np.float32(u1) * np.float32(hint - 1)))
        else:
            first = 2
        last = first + span_len
        src_mask[r, first:last] = True
        to_recover.append(src_text[r, first - 1:last])
        to_recover_pos.append(torch.arange(first - 1, min(last, width)))
    to_recover = pad_sequence(to_recover, batch_first=True, padding_value=pad_id)
    positions = pad_sequence(to_recover_pos, batch_first=True, padding_value=width - 1)
    targets = src_text[:, 1:][src_mask[:, 1:]]
    mask_idx = src_text[src_mask]
    repl = []
    for r, c in src_mask.nonzero().tolist():
        u, rr = draw_token(r, c)
        if u < np.float32(0.8):
            repl.append(mask_id)
        elif u < np.float32(0.9):
            repl.append(n_special + min(vocab - n_special - 1, int(np.float32(rr) * np.float32(vocab - n_special))))
        else:
            repl.append(int(src_text[r, c]))
    src_text[src_mask] = torch.tensor(repl, dtype=torch.long)
    return {"src_mask": src_mask, "targets": targets, "src_text": src_text, "to_recover": to_recover,
            "positions": positions, "mask_idx": mask_idx}
