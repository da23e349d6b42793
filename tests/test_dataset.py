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


# ------------------------------------------------------------------ the reference's own example files (round 3)
import json
import os

KAT = os.path.join(os.path.dirname(__file__), "golden", "marshal_kat")


def test_create_mt_batches_writes_what_the_reference_wrote(tmp_path):
    """tests/golden/marshal_kat/ holds what the REFERENCE's src/create_mt_batches.py write() produced (tokenizer trained and
    re-loaded by its own src/textprocessor.py) for a 50-line corpus: the product's writer, on the same corpus and the same
    tokenizer files, must produce equal example lists (parallel: sorted by target length, file order among ties, empty and
    out-of-range lines dropped; monolingual: ``<output>.0``)."""
    from imagetranslate_amd import create_mt_batches as C
    from imagetranslate_amd.textprocessor import TextProcessor
    meta = json.load(open(os.path.join(KAT, "meta.json")))
    tp = TextProcessor(os.path.join(KAT, "tok"))
    assert tp.tokenizer.get_vocab_size() == meta["vocab_size"]
    src_lang, dst_lang = tp.token_id("<en>"), tp.token_id("<fa>")
    assert (src_lang, dst_lang) == (meta["src_lang_id"], meta["dst_lang_id"])
    assert tp.tokenize_one_sentence_with_langid(meta["sample_line"], src_lang) == meta["sample_ids"]
    out = tmp_path / "mt.marshal"
    n = C.write(tp, str(out), os.path.join(KAT, "src.txt"), src_lang, os.path.join(KAT, "dst.txt"), dst_lang,
                min_len=meta["min_len_parallel"], max_len=meta["max_len_parallel"])
    ref = marshal.load(open(os.path.join(KAT, "mt.marshal"), "rb"))
    assert n == meta["n_parallel"] == len(ref) < 49  # the empty line and the over-long ones are gone
    assert marshal.load(open(out, "rb")) == ref  # (raw bytes differ in marshal's FLAG_REF bits, which follow refcounts at dump time)
    mono = tmp_path / "mass.marshal"
    n = C.write(tp, str(mono), os.path.join(KAT, "src.txt"), src_lang)
    ref_m = marshal.load(open(os.path.join(KAT, "mass.marshal.0"), "rb"))
    assert n == meta["n_mono"] == len(ref_m)
    assert marshal.load(open(str(mono) + ".0", "rb")) == ref_m


def test_datasets_read_the_reference_written_files():
    """MTDataset / MassDataset on the files the reference's writer produced: every example lands in exactly one batch, rows
    carry the language tag first and </s> last, pads are zeros (src/dataset.py:99-165,212-269 through the restatement)."""
    ref = marshal.load(open(os.path.join(KAT, "mt.marshal"), "rb"))
    ds = MTDataset(max_batch_capacity=1000, max_batch=120, pad_idx=0, batch_pickle_dir=os.path.join(KAT, "mt.marshal"))
    _same(ds.batches, DO.mt_batches(ref, 120, 1000, 175, 1, 0))
    rows = [(b["src_texts"][i][b["src_pad_mask"][i]].tolist(), b["dst_texts"][i][b["dst_pad_mask"][i]].tolist())
            for b in ds.batches for i in range(b["src_texts"].size(0))]
    assert sorted(rows) == sorted((e[0], e[1]) for e in ref)
    for s, d in rows:
        assert s[0] == 5 and d[0] == 6 and s[-1] == 4 and d[-1] == 4
    mono = marshal.load(open(os.path.join(KAT, "mass.marshal.0"), "rb"))
    md = MassDataset(batch_pickle_dir=os.path.join(KAT, "mass.marshal"), max_batch_capacity=1000, max_batch=200, pad_idx=0,
                     max_seq_len=175, keep_examples=False)
    got = sorted(b["src_texts"][i][b["src_texts"][i] != 0].tolist() for b in md.batches for i in range(b["src_texts"].size(0)))
    assert got == sorted(e[0] for e in mono)
