"""On-disk example files and batch construction -- drop-in for the text parts of the reference's ``src/dataset.py``
(``MTDataset`` ``:73-168``, ``MassDataset`` ``:171-270``) and the example files written by
``src/create_mt_batches.py:8-71``.

File format (unchanged, so files written by the reference load here and vice versa): a ``marshal`` dump of a list of
``(src_ids, dst_ids, src_lang, dst_lang)`` tuples sorted by target length (parallel data), or of
``(src_ids, lang)`` tuples sorted by length (monolingual data for MASS, possibly split into ``<path>.<part>`` files).

Batches are whole pre-padded tensors (the trainer's DataLoader uses ``batch_size=1``): sentences are taken in file
order and a batch is closed as soon as adding the next sentence would exceed either budget
    parallel : (S + T) * n > max_batch          or  (S^2 + T^2) * n * T > max_batch_capacity * 1e6
    MASS     : 2 * S * n   > max_batch          or  2 * S^3 * n         > max_batch_capacity * 1e6
(S, T = longest source / target in the batch including the candidate, n = sentences including the candidate), as
long as the batch without the candidate keeps at least ``ngpu`` sentences.  The shapes this produces (length-sorted,
ragged) are what the kernels see in real training.
"""
import glob
import marshal
from typing import List, Optional

import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import Dataset


def _first_pad_index(texts: torch.Tensor, pad_idx: int) -> torch.Tensor:
    """Per row: index of the first pad token, or width-1 when the row has none (reference ``pad_idx_find``)."""
    is_pad = texts == pad_idx
    width = texts.size(1)
    first = torch.where(is_pad.any(dim=1), is_pad.to(torch.int8).argmax(dim=1), torch.full((texts.size(0),), width - 1))
    return first.to(torch.long)


class MTDataset(Dataset):
    def __init__(self, max_batch_capacity: int, max_batch: int, pad_idx: int, max_seq_len: int = 175,
                 batch_pickle_dir: Optional[str] = None, examples: Optional[List] = None, lex_dict=None,
                 keep_pad_idx: bool = True, ngpu: int = 1):
        if lex_dict is not None:
            raise NotImplementedError("lexical proposals (--dict) are outside the hot path (SURVEY a5)")
        self.lex_dict = None
        self.keep_pad_idx = keep_pad_idx
        self.ngpu = ngpu
        if examples is None:
            with open(batch_pickle_dir, "rb") as fr:
                examples = marshal.load(fr)
        self.batch_examples(examples, max_batch, max_batch_capacity, max_seq_len, ngpu, pad_idx)

    def _emit(self, src, dst, src_langs, dst_langs, pad_idx):
        src_batch = pad_sequence(src, batch_first=True, padding_value=pad_idx)
        dst_batch = pad_sequence(dst, batch_first=True, padding_value=pad_idx)
        entry = {"src_texts": src_batch, "src_pad_mask": src_batch != pad_idx, "dst_texts": dst_batch,
                 "dst_pad_mask": dst_batch != pad_idx, "src_langs": torch.LongTensor(src_langs),
                 "dst_langs": torch.LongTensor(dst_langs), "proposal": torch.LongTensor([pad_idx])}
        if self.keep_pad_idx:
            entry["pad_idx"] = _first_pad_index(src_batch, pad_idx)
        self.batches.append(entry)

    def batch_examples(self, examples, max_batch, max_batch_capacity, max_seq_len, num_gpu, pad_idx):
        self.batches = []
        budget = max_batch_capacity * 1000000
        src, dst, sl, dl = [], [], [], []
        max_s = max_t = 0
        for ex in examples:
            s = torch.LongTensor(list(ex[0][:max_seq_len]))
            t = torch.LongTensor(list(ex[1][:max_seq_len]))
            new_s, new_t, n = max(max_s, s.numel()), max(max_t, t.numel()), len(src) + 1
            over = (new_s + new_t) * n > max_batch or (new_s ** 2 + new_t ** 2) * n * new_t > budget
            if over and len(src) >= num_gpu and n > 1:
                self._emit(src, dst, sl, dl, pad_idx)
                src, dst, sl, dl = [], [], [], []
                new_s, new_t = s.numel(), t.numel()
            src.append(s); dst.append(t); sl.append(ex[2]); dl.append(ex[3])
            max_s, max_t = new_s, new_t
        if len(src) > 0 and len(src) >= num_gpu:
            self._emit(src, dst, sl, dl, pad_idx)

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, item):
        return self.batches[item]


class MassDataset(Dataset):
    def __init__(self, batch_pickle_dir: Optional[str], max_batch_capacity: int, max_batch: int, pad_idx: int,
                 max_seq_len: int = 512, keep_examples: bool = False, example_list: Optional[List] = None, lex_dict=None,
                 keep_pad_idx: bool = True, ngpu: int = 1):
        if lex_dict is not None:
            raise NotImplementedError("lexical proposals (--dict) are outside the hot path (SURVEY a5)")
        self.lex_dict = None
        self.keep_pad_idx = keep_pad_idx
        self.ngpu = ngpu
        if example_list is None:
            self.examples_list = [self.read_example_file(path) for path in sorted(glob.glob(batch_pickle_dir + "*"))]
        else:
            self.examples_list = example_list
        self.batch_items(max_batch, max_batch_capacity, max_seq_len, pad_idx)
        if example_list is None and not keep_examples:
            self.examples_list = []

    @staticmethod
    def read_example_file(path):
        with open(path, "rb") as fr:
            return marshal.load(fr)

    def batch_items(self, max_batch, max_batch_capacity, max_seq_len, pad_idx):
        self.batches = []
        self.lang_ids = set()
        budget = max_batch_capacity * 1000000
        groups = []
        cur, langs, longest = [], [], 0
        for examples in self.examples_list:
            for ex in examples:
                if len(ex[0]) > max_seq_len:
                    continue
                ids = list(ex[0])
                self.lang_ids.add(int(ids[0]))
                new_longest, n = max(longest, len(ids)), len(cur) + 1
                over = 2 * new_longest * n > max_batch or 2 * new_longest ** 3 * n > budget
                if over and len(cur) >= self.ngpu and n > 1:
                    groups.append((cur, langs))
                    cur, langs, new_longest = [], [], len(ids)
                cur.append(ids); langs.append(ex[1])
                longest = new_longest
        if len(cur) > 0 and len(cur) >= self.ngpu:
            groups.append((cur, langs))
        for sents, lg in groups:
            texts = pad_sequence([torch.LongTensor(s) for s in sents], batch_first=True, padding_value=pad_idx)
            self.batches.append({"src_texts": texts, "langs": torch.LongTensor(lg), "proposal": torch.LongTensor([pad_idx]),
                                 "pad_idx": _first_pad_index(texts, pad_idx)})

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, item):
        return self.batches[item]
