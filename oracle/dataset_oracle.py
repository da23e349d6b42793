"""CPU oracle for batch construction -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, loop for loop, the packing rules of the reference's ``MTDataset.batch_examples`` (``src/dataset.py:99-165``)
and ``MassDataset.batch_items`` (``:212-269``) with ``lex_dict=None``: append the example, recompute the two budgets,
and when one is exceeded emit everything BUT the last example and restart the batch from it.  The reference's
``src/dataset.py`` cannot be imported here (it imports ``torchvision``, absent), so this is pinned by the text of the
reference only: "parity unpinned" beyond agreement between this loop form and the product's incremental form.
"""
import torch
from torch.nn.utils.rnn import pad_sequence


def _pad_indices(texts, pad_idx):
    out = []
    for row in texts == pad_idx:
        nz = torch.nonzero(row)
        out.append(int(row.numel()) - 1 if nz.size(0) == 0 else int(nz[0]))
    return torch.LongTensor(out)


def mt_batches(examples, max_batch, max_batch_capacity, max_seq_len, num_gpu, pad_idx, keep_pad_idx=True):
    batches = []
    cur_src, cur_dst, cur_sl, cur_dl = [], [], [], []
    max_s = max_t = 0

    def emit(src, dst, sl, dl):
        sb = pad_sequence(src, batch_first=True, padding_value=pad_idx)
        db = pad_sequence(dst, batch_first=True, padding_value=pad_idx)
        e = {"src_texts": sb, "src_pad_mask": sb != pad_idx, "dst_texts": db, "dst_pad_mask": db != pad_idx,
             "src_langs": torch.LongTensor(sl), "dst_langs": torch.LongTensor(dl), "proposal": torch.LongTensor([pad_idx])}
        if keep_pad_idx:
            e["pad_idx"] = _pad_indices(sb, pad_idx)
        batches.append(e)

    for ex in examples:
        src = torch.LongTensor(ex[0][:max_seq_len])
        dst = torch.LongTensor(ex[1][:max_seq_len])
        cur_sl.append(ex[2]); cur_dl.append(ex[3])
        max_s = max(max_s, int(src.size(0))); max_t = max(max_t, int(dst.size(0)))
        cur_src.append(src); cur_dst.append(dst)
        capacity = (max_s ** 2 + max_t ** 2) * len(cur_src) * max_t
        size = (max_s + max_t) * len(cur_src)
        if (size > max_batch or capacity > max_batch_capacity * 1000000) and len(cur_src[:-1]) >= num_gpu and len(cur_src) > 1:
            emit(cur_src[:-1], cur_dst[:-1], cur_sl[:-1], cur_dl[:-1])
            cur_src, cur_dst = [cur_src[-1]], [cur_dst[-1]]
            cur_sl, cur_dl = [cur_sl[-1]], [cur_dl[-1]]
            max_s, max_t = int(cur_src[0].size(0)), int(cur_dst[0].size(0))
    if len(cur_src) > 0 and len(cur_src) >= num_gpu:
        emit(cur_src, cur_dst, cur_sl, cur_dl)
    return batches


def mass_batches(examples_list, max_batch, max_batch_capacity, max_seq_len, ngpu, pad_idx):
    groups, langs_out = [], []
    cur, cur_langs, longest = [], [], 0
    for examples in examples_list:
        for ex in examples:
            if len(ex[0]) > max_seq_len:
                continue
            cur_langs.append(ex[1])
            longest = max(longest, len(ex[0]))
            cur.append(ex[0])
            capacity = 2 * (longest ** 3) * len(cur)
            size = 2 * longest * len(cur)
            if (size > max_batch or capacity > max_batch_capacity * 1000000) and len(cur[:-1]) >= ngpu and len(cur_langs) > 1:
                groups.append(cur[:-1]); langs_out.append(cur_langs[:-1])
                cur, cur_langs = [cur[-1]], [cur_langs[-1]]
                longest = len(cur[0])
    if len(cur) > 0 and len(cur) >= ngpu:
        groups.append(cur); langs_out.append(cur_langs)
    out = []
    for g, l in zip(groups, langs_out):
        texts = pad_sequence([torch.LongTensor(list(s)) for s in g], batch_first=True, padding_value=pad_idx)
        out.append({"src_texts": texts, "langs": torch.LongTensor(l), "proposal": torch.LongTensor([pad_idx]),
                    "pad_idx": _pad_indices(texts, pad_idx)})
    return out
