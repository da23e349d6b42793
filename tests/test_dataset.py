"""Host-side batch construction: the product's incremental packing against the loop-for-loop
restatement of the reference's rules (oracle/dataset_oracle.py), and the marshal example-file round trip."""
import marshal
import random

import pytest
import torch

from imagetranslate_amd.dataset import MassDataset, MTDataset
from oracle import dataset_oracle as DO


def _parallel_examples(n, seed, max_len=60):
    rnd = random.Random(seed)
    ex = []
    for _ in range(n):
        ls, lt = rnd.randint(2, max_len), rnd.randint(2, max_len)
        ex.append(([5] + [rnd.randint(7, 999) for _ in range(ls - 2)] + [4], [6] + [rnd.randint(7, 999) for _ in range(lt - 2)] + [4], 0, 1))
    ex.sort(key=lambda e: len(e[1]))
    return ex


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x.keys() == y.keys()
        for k in x:
            assert torch.equal(x[k], y[k]), k


@pytest.mark.parametrize("max_batch,capacity,ngpu,max_seq_len", [(400, 1, 1, 175), (2000, 40, 1, 175), (2000, 40, 4, 30), (100000, 2, 2, 175), (50, 1000, 1, 175)])
def test_mt_dataset_matches_reference_packing_rules(max_batch, capacity, ngpu, max_seq_len):
    ex = _parallel_examples(300, seed=max_batch + ngpu)
    ours = MTDataset(max_batch_capacity=capacity, max_batch=max_batch, pad_idx=0, max_seq_len=max_seq_len, examples=ex, ngpu=ngpu)
    ref = DO.mt_batches(ex, max_batch, capacity, max_seq_len, ngpu, 0)
    _same(ours.batches, ref)
    assert sum(b["src_texts"].size(0) for b in ours.batches) <= len(ex)
    for b in ours.batches:
        assert b["src_texts"].size(0) >= ngpu and b["src_texts"].size(1) <= max_seq_len
        assert torch.equal(b["src_pad_mask"], b["src_texts"] != 0)


def test_mt_dataset_reads_marshal_file(tmp_path):
    ex = _parallel_examples(40, seed=3)
    path = tmp_path / "train.batch"
    with open(path, "wb") as fw:
        marshal.dump(ex, fw)
    ours = MTDataset(max_batch_capacity=20, max_batch=600, pad_idx=0, batch_pickle_dir=str(path))
    _same(ours.batches, DO.mt_batches(ex, 600, 20, 175, 1, 0))
    assert len(ours) == len(ours.batches) and ours[0] is ours.batches[0]


@pytest.mark.parametrize("max_batch,capacity,ngpu", [(300, 1, 1), (4000, 5, 2), (100, 1000, 1)])
def test_mass_dataset_matches_reference_packing_rules(tmp_path, max_batch, capacity, ngpu):
    rnd = random.Random(11)
    parts = []
    for part in range(2):
        ex = [([5] + [rnd.randint(7, 999) for _ in range(rnd.randint(1, 70))] + [4], 0) for _ in range(120)]
        ex.sort(key=lambda e: len(e[0]))
        parts.append(ex)
        with open(str(tmp_path / "mono.batch") + "." + str(part), "wb") as fw:
            marshal.dump(ex, fw)
    ours = MassDataset(str(tmp_path / "mono.batch"), max_batch_capacity=capacity, max_batch=max_batch, pad_idx=0, max_seq_len=64, ngpu=ngpu)
    _same(ours.batches, DO.mass_batches(parts, max_batch, capacity, 64, ngpu, 0))
    assert ours.lang_ids == {5}
    for b in ours.batches:
        assert b["src_texts"].size(1) <= 64
