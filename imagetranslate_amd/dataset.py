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


# ------------------------------------------------------------------------------------------- image / caption data
# The CNN trunk is outside the hot path: what the reference computes from pixels with a frozen
# torchvision ResNet (``ModifiedResnet`` up to ``x8.view().permute()``, src/image_model.py:24-36) enters here as
# pre-extracted region features.  A feature directory holds ``features.pt`` = torch.save({"paths": [str, ...],
# "feats": Tensor[n_images, regions, C]}) (read with ``weights_only=True``).  The caption file is the reference's own:
# marshal of (unique_images: {image id: path}, captions: [(image id, caption ids), ...]) (src/dataset.py:301-306).
class RegionFeatures:
    def __init__(self, root_img_dir: str):
        import os
        blob = torch.load(os.path.join(root_img_dir, "features.pt"), map_location="cpu", weights_only=True)
        self.paths = list(blob["paths"])
        self.feats = blob["feats"]
        assert self.feats.dim() == 3 and self.feats.size(0) == len(self.paths), "features.pt: feats must be [n_images, regions, C]"
        self.index = {p: i for i, p in enumerate(self.paths)}

    def get(self, paths) -> torch.Tensor:
        """[len(paths), regions, C]; an image without features gets zeros (the reference substitutes a black image, :367)."""
        rows = [self.index.get(p, -1) for p in paths]
        out = self.feats[[max(r, 0) for r in rows]].clone()
        for k, r in enumerate(rows):
            if r < 0:
                out[k].zero_()
        return out


class ImageCaptionDataset(Dataset):
    """src/dataset.py:278-376 with region features in place of pixels: captions in file order, a batch closed when it
    would hold more than ``max_img_per_batch`` images or 2 * L^3 * n > max_capacity * 1e6 (L = longest caption)."""

    def __init__(self, root_img_dir: str, data_bin_file: str, max_capacity: int, text_processor, max_img_per_batch: int,
                 lex_dict=None, ngpu: int = 1, use_neg_samples: bool = False, features: Optional[RegionFeatures] = None):
        if lex_dict is not None:
            raise NotImplementedError("lexical proposals (--dict) are outside the hot path (SURVEY a5)")
        self.ngpu = ngpu
        self.pad_idx = text_processor.pad_token_id()
        self.features = features if features is not None else RegionFeatures(root_img_dir)
        self.batches, self.image_batches, self.all_captions, self.lang_ids = [], [], [], set()
        budget = max_capacity * 1000000
        with open(data_bin_file, "rb") as fp:
            self.unique_images, captions = marshal.load(fp)
        tag = text_processor.id2token(captions[0][1][0])
        self.lang_ids.add(int(captions[0][1][0]))
        self.lang = text_processor.languages[tag] if tag in text_processor.languages else 0
        cur, imgs, longest = [], [], 0
        for image_id, caption in captions:
            if str(self.unique_images[image_id]).lower().endswith(".png"):
                continue
            cap = torch.LongTensor(list(caption))
            self.all_captions.append(cap)
            cur.append(cap); imgs.append(image_id)
            longest = max(longest, cap.numel())
            over = len(imgs) > max_img_per_batch or 2 * longest ** 3 * len(cur) > budget
            if over and len(cur) - 1 >= self.ngpu and len(cur) > 1:
                self._emit(cur[:-1], imgs[:-1])
                cur, imgs = [cur[-1]], [imgs[-1]]
                longest = cur[0].numel()
        if cur:
            self._emit(cur, imgs)

    def _emit(self, caps, imgs):
        texts = pad_sequence(caps, batch_first=True, padding_value=self.pad_idx)
        self.batches.append((texts, texts != self.pad_idx, _first_pad_index(texts, self.pad_idx)))
        self.image_batches.append(list(imgs))

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, item):
        texts, mask, pad_indices = self.batches[item]
        feats = self.features.get([self.unique_images[i] for i in self.image_batches[item]])
        return {"images": feats, "captions": texts, "pad_idx": pad_indices, "langs": torch.LongTensor([self.lang] * texts.size(0)),
                "caption_mask": mask, "proposal": None}


class ImageCaptionTestDataset(ImageCaptionDataset):
    """src/dataset.py:396-421: one row per distinct image of the batch, its captions kept as references."""

    def __getitem__(self, item):
        texts, _, _ = self.batches[item]
        refs, order = {}, []
        for row, image_id in enumerate(self.image_batches[item]):
            if image_id not in refs:
                refs[image_id] = []
                order.append(image_id)
            cap = texts[row]
            refs[image_id].append(cap)
        max_len = int(texts.size(1))
        first_tokens = torch.LongTensor([int(refs[i][0][0]) for i in order])
        feats = self.features.get([self.unique_images[i] for i in order])
        return {"images": feats, "img_ids": order, "captions": refs, "first_tokens": first_tokens,
                "langs": torch.LongTensor([self.lang] * len(order)), "max_len": max_len + 10, "proposal": None}


class ImageDataset(Dataset):
    """src/dataset.py:424-476 (inference): every image of the feature directory, ``max_img_per_batch`` per batch."""

    def __init__(self, root_img_dir: str, max_img_per_batch: int, target_lang: int, first_token: int,
                 features: Optional[RegionFeatures] = None):
        self.target_lang, self.first_token = target_lang, first_token
        self.features = features if features is not None else RegionFeatures(root_img_dir)
        paths = [p for p in self.features.paths if not p.lower().endswith(".png")]
        self.image_batches = [paths[i:i + max_img_per_batch] for i in range(0, len(paths), max_img_per_batch)]

    def __len__(self):
        return len(self.image_batches)

    def __getitem__(self, item):
        paths = self.image_batches[item]
        return {"images": self.features.get(paths), "tgt_langs": torch.LongTensor([self.target_lang] * len(paths)),
                "first_tokens": torch.LongTensor([self.first_token] * len(paths)), "paths": paths}
