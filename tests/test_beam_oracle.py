"""CPU checks for beam search: the post-processing helper against the reference's own function (known-answer
vectors generated from /root/reference/src/seq_gen.py, tests/golden/beam_kat.json), and the oracle BeamDecoder
against the committed token ids (drift guard; beam search itself is parity-unpinned in the reference)."""
import json
import os

import pytest
import torch

from oracle import reference_model as R
from oracle import seq_gen as OG
from imagetranslate_amd import seq_gen as PG
from util import beam_inputs, beam_state_dict, caption_beam_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("impl", [OG.get_outputs_until_eos, PG.get_outputs_until_eos], ids=["oracle", "product"])
def test_get_outputs_until_eos_matches_reference_vectors(impl):
    kat = json.load(open(os.path.join(GOLD, "beam_kat.json")))
    for c in kat["cases"]:
        lim = None if c["size_limit"] is None else torch.tensor(c["size_limit"])
        got = impl(c["eos"], torch.tensor(c["outputs"]), size_limit=lim, remove_first_token=c["remove_first_token"])
        assert [g.tolist() for g in got] == c["expected"]


def _toy_model():
    fx = torch.load(os.path.join(GOLD, "toy_seq2seq.pt"), weights_only=True)
    tp = R.SyntheticTextProcessor(1000)
    m = R.Seq2Seq(tp, lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512, num_attention_heads=4)
    return m, fx["state_dict"]


def test_oracle_beam_matches_fixture():
    gold = torch.load(os.path.join(GOLD, "toy_beam.pt"), weights_only=True)
    m, sd = _toy_model()
    m.load_state_dict(beam_state_dict(sd))
    m.eval()
    for name, kw in [("beam4", dict(beam_width=4)), ("beam1", dict(beam_width=1)),
                     ("beam3_padded", dict(beam_width=3, unpad_output=False)), ("beam4_maxlen10", dict(beam_width=4, max_len=10))]:
        out = OG.BeamDecoder(m, beam_width=5)(pad_idx=0, **beam_inputs(), **kw)
        assert [o.tolist() for o in out] == [t.tolist() for t in gold[name]["tokens"]], name
    # finished hypotheses really occur (EOS / PAD bookkeeping exercised), and so does the length limit
    lens = [len(t) for t in gold["beam4"]["tokens"]]
    assert min(lens) < max(lens)


def test_oracle_beam_properties():
    """Size-independent properties: beam 1 == greedy argmax decoding; no EOS inside an unpadded output; every
    output starts with the first token and respects the per-sentence length limit."""
    m, sd = _toy_model()
    m.load_state_dict(beam_state_dict(sd))
    m.eval()
    inp = beam_inputs()
    bd = OG.BeamDecoder(m, beam_width=1)
    out = bd(pad_idx=0, **inp)
    max_lens = [min(int(1.1 * int(s) + 5), 512) for s in inp["src_sizes"]]
    for b, o in enumerate(out):
        assert int(o[0]) == 5 and (o != 4).all() and len(o) <= max_lens[b]
    # greedy re-derivation of sentence 0 with the plain model forward
    b = 0
    src, mask = inp["src_inputs"][b:b + 1], inp["src_mask"][b:b + 1]
    enc = m.encode(src, mask, inp["src_langs"][b:b + 1].unsqueeze(-1).expand(-1, src.size(1)))[0]
    toks = [5]
    with torch.no_grad():
        for _ in range(len(out[b]) - 1):
            ids = torch.tensor([toks])
            h = m.decoder(encoder_states=enc, input_ids=ids, encoder_attention_mask=mask,
                          tgt_attention_mask=torch.ones_like(ids), token_type_ids=torch.ones_like(ids))[:, -1]
            toks.append(int(m.output_layer[1](h).argmax(-1)))
    assert toks == out[b].tolist()


def test_oracle_caption_beam_matches_fixture():
    gold = torch.load(os.path.join(GOLD, "toy_beam.pt"), weights_only=True)
    _, sd = _toy_model()
    tp = R.SyntheticTextProcessor(1000)
    cap = R.ImageCaptioning(tp, lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512,
                            num_attention_heads=4, image_feat_dim=64)
    cap.load_state_dict({**beam_state_dict(sd), **gold["caption_beam3"]["extra_state"]})
    cap.eval()
    out = OG.BeamDecoder(cap, beam_width=3)(pad_idx=0, max_len=14, **caption_beam_inputs())
    assert [o.tolist() for o in out] == [t.tolist() for t in gold["caption_beam3"]["tokens"]]
