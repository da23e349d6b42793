"""MASS batch construction on the device against the loop-form oracle on the same counter-based
draws (integer outputs: bit-exact), its statistics against the reference's 20/20/60 and 80/10/10 rules, and the
in-place unmask round trip."""
import pytest
import torch

from oracle import batch_oracle as BO
from oracle.reference_model import SyntheticTextProcessor

pytestmark = pytest.mark.gpu


def _batch(n_rows, width, seed, vocab=500):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(6, width + 1, (n_rows,), generator=g)
    text = torch.randint(7, vocab, (n_rows, width), generator=g)
    text[:, 0] = 5
    text[torch.arange(width)[None] >= lens[:, None]] = 0
    pad_idx = torch.where(lens < width, lens, torch.full_like(lens, width - 1))
    return text, pad_idx


@pytest.mark.parametrize("n_rows,width,seed", [(7, 24, 1), (64, 128, 2), (3, 9, 3), (200, 61, 4)])
def test_mass_mask_device_matches_oracle(cuda, n_rows, width, seed):
    from imagetranslate_amd.utils import mass_mask_device, mass_unmask_device
    tp = SyntheticTextProcessor(500)
    text, pad_idx = _batch(n_rows, width, seed)
    n_special = len(tp.special_tokens)
    exp = BO.mass_mask_reference(0.3, pad_idx, text, n_special, 500, 3, 0,
                                 lambda r: (BO.counter_uniform(seed, 0, r), BO.counter_uniform(seed, 1, r)),
                                 lambda r, c: (BO.counter_uniform(seed, 2, r * width + c), BO.counter_uniform(seed, 3, r * width + c)))
    dev_text = text.clone().cuda()
    got = mass_mask_device(0.3, pad_idx, dev_text, tp, seed)
    for k in ("src_mask", "targets", "src_text", "to_recover", "positions", "mask_idx"):
        assert torch.equal(got[k].cpu(), exp[k]), k
    assert got["src_text"].data_ptr() == dev_text.data_ptr(), "masked in place like the reference"
    mass_unmask_device(got)
    assert torch.equal(dev_text.cpu(), text), "unmask restores the batch"


def test_mass_mask_device_statistics(cuda):
    from imagetranslate_amd.utils import mass_mask_device
    tp = SyntheticTextProcessor(500)
    text, pad_idx = _batch(4000, 64, 9)
    got = mass_mask_device(0.3, pad_idx, text.clone().cuda(), tp, 77)
    mask = got["src_mask"].cpu()
    first = mask.to(torch.int8).argmax(1)
    hint = torch.ceil(pad_idx.float() - (1 - 0.3) * pad_idx.float()).long()
    assert torch.equal(mask.sum(1), pad_idx // 2), "span length is int(pad_index / 2)"
    assert (first >= 1).all() and (first <= torch.maximum(hint, torch.full_like(hint, 2))).all()
    assert abs(float((first == 1).float().mean()) - 0.2) < 0.03
    masked_new, masked_old = got["src_text"].cpu()[mask], text[mask]
    frac_mask = float((masked_new == 3).float().mean())
    frac_same = float((masked_new == masked_old).float().mean())
    assert abs(frac_mask - 0.8) < 0.01 and abs(frac_same - 0.1) < 0.01
    rnd = masked_new[(masked_new != 3) & (masked_new != masked_old)]
    assert int(rnd.min()) >= len(tp.special_tokens) and int(rnd.max()) < 500


@pytest.mark.parametrize("B,T,p", [(64, 128, 1.0), (64, 128, 0.7), (3, 5, 0.5), (1, 2, 1.0), (7, 300, 0.0), (40, 129, 0.93)])
def test_select_plan_matches_boolean_indexing(cuda, B, T, p):
    """imt_select_plan == (nonzero(mask[:, 1:]), ids[:, 1:][mask[:, 1:]]) of src/seq2seq.py:175-177 -- bit-exact."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(B * 1000 + T)
    ids = torch.randint(0, 30000, (B, T), generator=g)
    mask = torch.rand(B, T, generator=g) < p
    idx, tg = O.select_plan(mask.cuda(), ids.cuda(), col0=1)
    exp_idx = torch.nonzero(mask[:, 1:].reshape(-1)).view(-1)
    assert idx.dtype == torch.int32 and torch.equal(idx.cpu().long(), exp_idx)
    assert torch.equal(tg.cpu(), ids[:, 1:][mask[:, 1:]])
    # uint8 mask and a strided (sliced) ids view
    wide = torch.randint(0, 30000, (B, T + 3), generator=g)
    idx2, tg2 = O.select_plan(mask.to(torch.uint8).cuda(), wide.cuda()[:, :T], col0=1)
    assert torch.equal(idx2.cpu().long(), exp_idx) and torch.equal(tg2.cpu(), wide[:, :T][:, 1:][mask[:, 1:]])
