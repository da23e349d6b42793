"""GPU parity of the incremental-decoding kernels and of BeamDecoder (drop-in for src/seq_gen.py) against the CPU
oracle (oracle/seq_gen.py) on the same weights and inputs.  Token ids must be bit-exact in fp32 compute mode."""
import ctypes
import math
import os

import pytest
import torch

from oracle import reference_model as R
from oracle import seq_gen as OG
from tests.util import assert_close, beam_inputs, beam_state_dict, caption_beam_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------------------------------------ decode attention
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dh,H,n_keys", [(64, 8, 37), (32, 4, 5), (64, 2, 130), (32, 8, 64), (64, 4, 1)])
def test_attention_decode_slots(cuda, dtype, dh, H, n_keys):
    import imagetranslate_amd.hip_ops as O
    g = torch.Generator().manual_seed(dh + n_keys)
    d, R_, r_max, t_max = dh * H, 12, 16, n_keys + 3
    cache = (torch.randn(r_max, t_max, 3 * d, generator=g) * 0.7).to(dtype).cuda()
    slots = torch.randint(0, r_max, (R_, t_max), generator=g, dtype=torch.int32).cuda()
    pos = n_keys - 1
    slots[:, pos] = torch.arange(R_, dtype=torch.int32)
    q = cache[:R_, pos, :d]
    out = O.attention_decode(q, cache[0, 0, d:], cache[0, 0, 2 * d:], n_keys, H, ld_row=t_max * 3 * d, ld_pos=3 * d,
                             slots=slots, ldq=t_max * 3 * d, rows=R_)
    c = cache.float().cpu()
    sl = slots.cpu().long()
    ref = torch.empty(R_, d)
    for r in range(R_):
        rows = sl[r, :n_keys]
        k = c[rows, torch.arange(n_keys), d:2 * d].view(n_keys, H, dh)
        v = c[rows, torch.arange(n_keys), 2 * d:].view(n_keys, H, dh)
        qq = c[r, pos, :d].view(H, dh)
        s = torch.einsum("hd,khd->hk", qq, k) / math.sqrt(dh)
        ref[r] = torch.einsum("hk,khd->hd", torch.softmax(s, -1), v).reshape(d)
    assert_close(out.float().cpu(), ref, 1e-5 if dtype == torch.float32 else 1.5e-2, "decode attention (slots)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_decode_cross_mask(cuda, dtype):
    import imagetranslate_amd.hip_ops as O
    g = torch.Generator().manual_seed(5)
    B, rep, Tk, H, dh = 3, 4, 21, 4, 32
    d = H * dh
    kv = torch.randn(B, Tk, 2 * d, generator=g).to(dtype).cuda()
    q = torch.randn(B * rep, d, generator=g).to(dtype).cuda()
    mask = (torch.arange(Tk)[None] < torch.tensor([21, 9, 1])[:, None])
    mask[2, :] = False  # fully masked sentence: -10000 on every key -> plain softmax of the raw scores
    out = O.attention_decode(q, kv[0, 0, :d], kv[0, 0, d:], Tk, H, ld_row=Tk * 2 * d, ld_pos=2 * d, rep=rep,
                             key_mask=mask.to(torch.uint8).cuda())
    k = kv.float().cpu()[:, :, :d].view(B, Tk, H, dh).repeat_interleave(rep, 0)
    v = kv.float().cpu()[:, :, d:].view(B, Tk, H, dh).repeat_interleave(rep, 0)
    s = torch.einsum("rhd,rkhd->rhk", q.float().cpu().view(-1, H, dh), k) / math.sqrt(dh)
    s = s + (1.0 - mask.float().repeat_interleave(rep, 0))[:, None, :] * -10000.0
    ref = torch.einsum("rhk,rkhd->rhd", torch.softmax(s, -1), v).reshape(-1, d)
    assert_close(out.float().cpu(), ref, 1e-5 if dtype == torch.float32 else 1.5e-2, "decode attention (cross)")


def test_attention_decode_rejects_bad_args(cuda):
    import imagetranslate_amd.hip_ops as O
    from imagetranslate_amd._lib import ImtError
    q = torch.zeros(4, 48, device="cuda")
    with pytest.raises(ImtError):
        O.attention_decode(q, q, q, 1, 2, ld_row=48, ld_pos=48)  # head_dim 24


# ------------------------------------------------------------------------------------------------ beam step
def _ref_beam_step(logits, scores, sizes, eos_in, max_lens, hist, step, B, beam, rep, V, ratio, pad, eos):
    """The reference's step (src/seq_gen.py:193-227) on CPU tensors with the oracle's stable top-k."""
    lp = torch.log_softmax(logits, -1)
    over = (max_lens < step + 1)
    lp[eos_in.bool()] = 0
    if step > 1:
        lp[over.repeat_interleave(rep)] = 0
    total = scores.unsqueeze(-1) + lp
    if beam > 1:
        total = total / torch.pow((sizes + 6.0) / 6.0, ratio).unsqueeze(-1)
    top, idx = OG.stable_topk(total.view(B, -1), beam)
    if step > 1:
        idx[over] = pad
        flat = idx.view(-1)
        flat[eos_in.bool()] = pad
        parent = idx // V
    else:
        parent = torch.zeros_like(idx)
    word = idx % V
    prow = (torch.arange(B)[:, None] * rep + parent).view(-1)
    new_hist = torch.cat([hist[prow, :step], word.view(-1, 1)], 1)
    new_sizes = sizes[prow] + (word.view(-1) != pad)
    new_eos = (new_hist == eos).any(1)
    return top.view(-1), new_sizes, new_eos, new_hist, prow


def _call_beam_step(logits, scores, sizes, eos_in, max_lens, hist, slots_in, step, B, beam, rep, V, t_max, ratio, pad, eos):
    """One imt_beam_step through the C ABI on copies of the given CPU tensors; returns the output buffers."""
    import imagetranslate_amd.hip_ops as O
    from imagetranslate_amd import _lib as L
    dev = "cuda"
    rows, r_out = B * rep, B * beam
    z = lambda *s, dtype: torch.zeros(*s, dtype=dtype, device=dev)
    d_logits, d_scores, d_sizes = logits.cuda(), scores.cuda(), sizes.cuda()
    d_eos, d_max, d_hist = eos_in.to(torch.uint8).cuda(), max_lens.cuda(), hist.cuda()
    d_slots = slots_in.cuda()
    cs, ci = z(rows, beam, dtype=torch.float32), z(rows, beam, dtype=torch.int32)
    o_scores, o_sizes, o_eos = z(r_out, dtype=torch.float32), z(r_out, dtype=torch.float32), z(r_out, dtype=torch.uint8)
    o_hist, o_slots = z(r_out, t_max, dtype=torch.int64), z(r_out, t_max, dtype=torch.int32)
    o_parent, o_tok, cnt = z(r_out, dtype=torch.int32), z(r_out, dtype=torch.int64), z(t_max, dtype=torch.int32)
    a = L.BeamArgs()
    a.B, a.beam, a.rep, a.V, a.step, a.t_max = B, beam, rep, V, step, t_max
    a.logits, a.ld = d_logits.data_ptr(), V
    a.scores_in, a.sizes_in, a.eos_in = d_scores.data_ptr(), d_sizes.data_ptr(), d_eos.data_ptr()
    a.max_lens, a.hist_in, a.slots_in = d_max.data_ptr(), d_hist.data_ptr(), d_slots.data_ptr()
    a.len_penalty_ratio, a.pad_idx, a.eos = ratio, pad, eos
    a.cand_scores, a.cand_idx = cs.data_ptr(), ci.data_ptr()
    a.scores_out, a.sizes_out, a.eos_out = o_scores.data_ptr(), o_sizes.data_ptr(), o_eos.data_ptr()
    a.hist_out, a.slots_out, a.parent_out, a.tokens_out = o_hist.data_ptr(), o_slots.data_ptr(), o_parent.data_ptr(), o_tok.data_ptr()
    a.eos_count = cnt.data_ptr()
    O.beam_step(a)
    torch.cuda.synchronize()
    return dict(scores=o_scores.cpu(), sizes=o_sizes.cpu(), eos=o_eos.cpu(), hist=o_hist.cpu(), slots=o_slots.cpu(),
                parent=o_parent.cpu(), tokens=o_tok.cpu(), eos_count=cnt.cpu(), cand_idx=ci.cpu(), cand_scores=cs.cpu())


@pytest.mark.parametrize("beam,step", [(4, 1), (4, 3), (1, 1), (1, 4), (7, 2)])
def test_beam_step_matches_reference_step(cuda, beam, step):
    g = torch.Generator().manual_seed(beam * 10 + step)
    B, V, t_max, pad, eos, ratio = 5, 300, 8, 0, 4, 0.8
    rep = 1 if step == 1 else beam
    rows = B * rep
    logits = torch.randn(rows, V, generator=g) * 3
    logits[0, 10] = logits[0, 20] = logits[0].max() + 1.0       # exact tie inside a row
    if rows > 2:
        logits[2] = logits[1]                                     # identical rows -> ties across rows
    hist = torch.randint(6, V, (rows, t_max), generator=g)
    hist[:, step:] = 0
    eos_in = torch.zeros(rows, dtype=torch.bool)
    if step > 1:
        eos_in[torch.randperm(rows, generator=g)[:rows // 3]] = True
        for r in torch.nonzero(eos_in).view(-1):
            hist[r, step - 1] = eos
    scores = -torch.rand(rows, generator=g) * 5
    if rows > 2:
        scores[2] = scores[1]
    sizes = torch.randint(1, step + 1, (rows,), generator=g).float()
    if rows > 2:
        sizes[2] = sizes[1]
    max_lens = torch.tensor([step + 1, step, 9, 9, step - 1])   # sentences 1 and 4 over the limit when step > 1
    exp = _ref_beam_step(logits.clone(), scores, sizes, eos_in, max_lens, hist, step, B, beam, rep, V, ratio, pad, eos)

    r_out = B * beam
    slots_in = torch.randint(0, r_out, (rows, t_max), generator=g, dtype=torch.int32)
    o = _call_beam_step(logits, scores, sizes, eos_in, max_lens, hist, slots_in, step, B, beam, rep, V, t_max, ratio, pad, eos)
    o_hist, o_parent, o_tok, o_eos, cnt = o["hist"], o["parent"], o["tokens"], o["eos"], o["eos_count"]
    o_scores, o_sizes, o_slots = o["scores"], o["sizes"], o["slots"]
    top, new_sizes, new_eos, new_hist, prow = exp
    assert torch.equal(o_hist.cpu()[:, :step + 1], new_hist), "token history must be bit-exact"
    assert torch.equal(o_parent.cpu().long(), prow)
    assert torch.equal(o_tok.cpu(), new_hist[:, step])
    assert torch.equal(o_eos.cpu().bool(), new_eos)
    assert int(cnt[step]) == int(new_eos.sum())
    assert_close(o_scores.cpu(), top, 1e-5, "beam scores")
    if beam > 1:
        assert torch.equal(o_sizes.cpu(), new_sizes)
    # slot table: ancestors' rows for positions < step, own row at `step`
    assert torch.equal(o_slots.cpu()[:, :step], slots_in.cpu()[prow, :step])
    assert torch.equal(o_slots.cpu()[:, step], torch.arange(r_out, dtype=torch.int32))


@pytest.mark.parametrize("beam,step", [(4, 1), (4, 3), (1, 2), (7, 2)])
def test_beam_step_with_nan_logits_stays_in_bounds(cuda, beam, step):
    """NaN logits (bf16 overflow, a damaged checkpoint) have no defined hypothesis, but every index the step writes
    must stay inside its buffer (decode.hip: the 0x7fffffff sentinels of the row top-k and of the merge), and
    sentences whose rows are finite must come out exactly as without the damage."""
    g = torch.Generator().manual_seed(1000 + beam * 10 + step)
    B, V, t_max, pad, eos, ratio = 6, 300, 8, 0, 4, 0.8
    rep = 1 if step == 1 else beam
    rows, r_out = B * rep, B * beam
    logits = torch.randn(rows, V, generator=g) * 3
    hist = torch.randint(6, V, (rows, t_max), generator=g)
    hist[:, step:] = 0
    eos_in = torch.zeros(rows, dtype=torch.bool)
    scores = -torch.rand(rows, generator=g) * 5
    sizes = torch.randint(1, step + 1, (rows,), generator=g).float()
    max_lens = torch.full((B,), 9)
    slots_in = torch.randint(0, r_out, (rows, t_max), generator=g, dtype=torch.int32)
    clean = _call_beam_step(logits, scores, sizes, eos_in, max_lens, hist, slots_in, step, B, beam, rep, V, t_max, ratio, pad, eos)

    bad = logits.clone()
    bad[0 * rep:1 * rep] = float("nan")                        # sentence 0: every row all-NaN
    bad[2 * rep] = float("nan")                                # sentence 2: its first row all-NaN, the others finite
    bad[3 * rep, 5::2] = float("nan")                          # sentence 3: NaN sprinkled inside a row
    bad_scores = scores.clone()
    bad_scores[4 * rep] = float("nan")                         # sentence 4: NaN carried in from the previous step
    o = _call_beam_step(bad, bad_scores, sizes, eos_in, max_lens, hist, slots_in, step, B, beam, rep, V, t_max, ratio, pad, eos)

    sent = torch.arange(r_out) // beam
    lo = sent * rep
    assert bool(((o["cand_idx"] >= 0) & (o["cand_idx"] < V)).all()), "row candidates must be vocabulary indices"
    assert bool(((o["parent"] >= lo) & (o["parent"] < lo + rep)).all()), "parents must be rows of the same sentence"
    assert bool(((o["tokens"] >= 0) & (o["tokens"] < V)).all())
    assert bool(((o["hist"][:, :step + 1] >= 0) & (o["hist"][:, :step + 1] < V)).all())
    assert torch.equal(o["slots"][:, step], torch.arange(r_out, dtype=torch.int32))
    assert 0 <= int(o["eos_count"][step]) <= r_out
    # the untouched sentences (1 and 5) are bit-identical to the clean run
    for b in (1, 5):
        sl = slice(b * beam, (b + 1) * beam)
        for k in ("scores", "sizes", "eos", "hist", "slots", "parent", "tokens"):
            assert torch.equal(o[k][sl], clean[k][sl]), (b, k)


# ------------------------------------------------------------------------------------------------ whole search
def _pair(cls="Seq2Seq", state=None, **kw):
    import imagetranslate_amd.image_model as I
    import imagetranslate_amd.seq2seq as S
    tp = R.SyntheticTextProcessor(1000)
    args = dict(lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512, num_attention_heads=4, **kw)
    ref = getattr(R, cls)(tp, **args)
    ref.load_state_dict(state)
    ours = {"Seq2Seq": S.Seq2Seq, "ImageCaptioning": I.ImageCaptioning}[cls](tp, **args)
    ours.load_state_dict(ref.state_dict(), strict=False)
    return ref.eval(), ours.cuda().eval()


def _fixture_state(*a):
    fx = torch.load(os.path.join(GOLD, "toy_seq2seq.pt"), weights_only=True)
    return beam_state_dict(fx["state_dict"], *a)


@pytest.mark.parametrize("kv_cache", [True, False], ids=["kv_cache", "recompute"])
def test_beam_decoder_fp32_tokens_bit_exact(cuda, kv_cache):
    from imagetranslate_amd.seq_gen import BeamDecoder
    gold = torch.load(os.path.join(GOLD, "toy_beam.pt"), weights_only=True)
    ref, ours = _pair(state=_fixture_state())
    inp = beam_inputs()
    for name, kw in [("beam4", dict(beam_width=4)), ("beam1", dict(beam_width=1)),
                     ("beam3_padded", dict(beam_width=3, unpad_output=False)), ("beam4_maxlen10", dict(beam_width=4, max_len=10))]:
        exp = OG.BeamDecoder(ref, beam_width=5)(pad_idx=0, **inp, **kw)
        got = BeamDecoder(ours, beam_width=5, kv_cache=kv_cache, sync_every=3)(pad_idx=0, **inp, **kw)
        assert [g.tolist() for g in got] == [e.tolist() for e in exp], name
        assert [g.tolist() for g in got] == [t.tolist() for t in gold[name]["tokens"]], name + " (fixture)"


def test_beam_decoder_soft_distribution_and_sync_period(cuda):
    """Flatter next-token distributions (closer scores) and different stop-check periods give the same tokens."""
    from imagetranslate_amd.seq_gen import BeamDecoder
    gold = torch.load(os.path.join(GOLD, "toy_beam.pt"), weights_only=True)
    ref, ours = _pair(state=_fixture_state(2.5, 2.5))
    inp = beam_inputs()
    exp = [t.tolist() for t in gold["beam4_soft"]["tokens"]]
    for sync in (1, 4, 100):
        got = BeamDecoder(ours, beam_width=4, sync_every=sync)(pad_idx=0, **inp)
        assert [g.tolist() for g in got] == exp
    got = BeamDecoder(ours, beam_width=4, sync_every=100)(pad_idx=0, unpad_output=False, **inp)
    exp_p = OG.BeamDecoder(ref, beam_width=4)(pad_idx=0, unpad_output=False, **inp)
    assert [g.tolist() for g in got] == [e.tolist() for e in exp_p], "padded outputs must stop at the reference's step"


def test_beam_decoder_caption(cuda):
    from imagetranslate_amd.seq_gen import BeamDecoder
    gold = torch.load(os.path.join(GOLD, "toy_beam.pt"), weights_only=True)
    state = {**_fixture_state(), **gold["caption_beam3"]["extra_state"]}
    ref, ours = _pair("ImageCaptioning", state=state, image_feat_dim=64)
    for kv in (True, False):
        got = BeamDecoder(ours, beam_width=3, kv_cache=kv)(pad_idx=0, max_len=14, **caption_beam_inputs())
        assert [g.tolist() for g in got] == [t.tolist() for t in gold["caption_beam3"]["tokens"]]
    # precomputed image embeddings (image_embed=) take the same path (src/seq_gen.py:100-103)
    emb = ours.encode(images=caption_beam_inputs()["images"].cuda())[0]
    inp = caption_beam_inputs()
    got = BeamDecoder(ours, beam_width=3)(pad_idx=0, max_len=14, image_embed=emb, first_tokens=inp["first_tokens"],
                                          tgt_langs=inp["tgt_langs"])
    assert [g.tolist() for g in got] == [t.tolist() for t in gold["caption_beam3"]["tokens"]]


def test_beam_decoder_bf16_and_list_wrapped_args(cuda):
    """bf16 compute mode: cached and recomputed decoding agree with each other on most sentences (bf16 rounding of
    the deliberately sharp toy weights flips a few near-ties, so this is a sanity bound, not a parity claim); arguments
    wrapped in 1-element lists (the reference's threaded-DP convention, src/seq_gen.py:57-71) are unwrapped."""
    from imagetranslate_amd.seq_gen import BeamDecoder
    ref, ours = _pair(state=_fixture_state())
    ours.set_compute_dtype(torch.bfloat16)
    inp = beam_inputs()
    a = BeamDecoder(ours, beam_width=4, kv_cache=True)(pad_idx=0, **{k: [v] for k, v in inp.items()})
    b = BeamDecoder(ours, beam_width=4, kv_cache=False)(pad_idx=0, **inp)
    exp = OG.BeamDecoder(ref, beam_width=4)(pad_idx=0, **inp)
    same = sum(int(x.tolist() == y.tolist()) for x, y in zip(a, b))
    assert same >= len(a) // 2, "bf16 cached vs recomputed decoding diverged on %d sentences" % (len(a) - same)
    agree = sum(int(x.tolist() == y.tolist()) for x, y in zip(a, exp))
    assert agree >= len(a) // 2, "bf16 beam search drifted far from the fp32 oracle"
    for o in a:
        assert int(o[0]) == 5 and (o != 4).all()


@pytest.mark.parametrize("beam", [1, 5])
def test_beam_search_with_the_one_launch_step(cuda, monkeypatch, beam):
    """A whole BeamDecoder search at the size where the one-launch decoder step applies (bf16, hidden size 512): the tokens of its first
    steps agree with the launch-per-operator chain on (nearly) every sentence -- later positions may part where bf16 rounding flips a
    near-tie of these random weights -- every output row is well-formed, and the end-of-search imt_decode_check raises nothing."""
    import imagetranslate_amd.seq2seq as S
    from imagetranslate_amd.seq_gen import BeamDecoder
    torch.manual_seed(11)
    tp = R.SyntheticTextProcessor(1000)
    ours = S.Seq2Seq(tp, lang_dec=False, enc_layer=2, dec_layer=3, embed_dim=512, intermediate_dim=2048, num_attention_heads=8).cuda().eval()
    ours.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(3)
    B, Sx = 12, 24
    src = torch.randint(6, 1000, (B, Sx), generator=g)
    src[:, 0], src[:, -1] = 5, 4
    mask = torch.ones(B, Sx, dtype=torch.bool)
    args = dict(src_inputs=src.cuda(), src_sizes=torch.full((B,), Sx), first_tokens=torch.full((B,), 5), src_mask=mask.cuda(),
                src_langs=torch.zeros(B, dtype=torch.long).cuda(), tgt_langs=torch.ones(B, dtype=torch.long).cuda(), pad_idx=0)
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("IMT_DECODE_FUSED", fused)
        outs[fused] = BeamDecoder(ours, beam_width=beam, kv_cache=True)(max_len=12, **args)
    head = sum(int(x[:4].tolist() == y[:4].tolist()) for x, y in zip(outs["1"], outs["0"]))
    assert head >= B - 2, "the first tokens of %d of %d sentences differ between the one-launch step and the chain" % (B - head, B)
    for o in outs["1"]:
        assert int(o[0]) == 5 and 2 <= len(o) <= 12 and bool(((o >= 0) & (o < 1000)).all())


def test_decode_step_matches_full_decoder(cuda):
    """imt_decode_step hidden states == last row of the full decoder forward on the same prefix (fp32)."""
    from imagetranslate_amd import _lib as L
    from imagetranslate_amd.param_store import store_of
    from imagetranslate_amd.seq_gen import _Incremental
    ref, ours = _pair(state=_fixture_state())
    inp = beam_inputs()
    B, S = inp["src_inputs"].shape
    enc = ours.encode(inp["src_inputs"], inp["src_mask"].cuda(), inp["src_langs"].unsqueeze(-1).expand(-1, S))[0].contiguous()
    T = 7
    g = torch.Generator().manual_seed(0)
    toks = torch.randint(6, 1000, (B, T), generator=g).cuda()
    types = torch.ones(B, T, dtype=torch.long).cuda()
    full = ours.decoder(encoder_states=enc, input_ids=toks, encoder_attention_mask=inp["src_mask"].cuda(),
                        tgt_attention_mask=torch.ones_like(toks), token_type_ids=types)
    store = store_of(ours.decoder).ensure()
    flat = store.params_for(torch.float32)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    inc = _Incremental(L.load(), ours.decoder, store, torch.float32, flat, enc, inp["src_mask"].to(torch.uint8).cuda(), B, 1, T, st)
    slots = torch.arange(B, dtype=torch.int32).unsqueeze(1).expand(B, T).contiguous().cuda()
    out = torch.empty(B, 128, device="cuda")
    for t in range(T):
        inc.step(t, B, 1, toks[:, t].contiguous(), types[:, 0].contiguous(), slots, out)
        assert_close(out, full[:, t], 1e-4, "decode step %d" % t)


@pytest.mark.parametrize("name", ["text_beam1", "text_beam4", "text_beam5", "text_beam4_eos_first", "caption_beam3"])
@pytest.mark.parametrize("kv_cache", [True, False], ids=["kv_cache", "recompute"])
def test_beam_first_step_against_reference_beam_decoder(cuda, name, kv_cache):
    """The product against what the REFERENCE's own src/seq_gen.py BeamDecoder.forward returned when run un-modified with
    max_len=2 on the oracle model (tests/golden/beam_step1_kat.pt): encode, first-token handling, log-softmax, length
    penalty and the top-k over V of step 1, bit-exact token ids (fp32 compute)."""
    from imagetranslate_amd.seq_gen import BeamDecoder
    from tests.test_oracle_pinning import check_beam_step1
    kat = torch.load(os.path.join(GOLD, "beam_step1_kat.pt"), weights_only=True)
    case = kat[name]
    if name.startswith("caption"):
        gold = torch.load(os.path.join(GOLD, "toy_beam.pt"), weights_only=True)
        _, ours = _pair("ImageCaptioning", state={**_fixture_state(), **gold["caption_beam3"]["extra_state"]}, image_feat_dim=64)
        inp = caption_beam_inputs()
    else:
        _, ours = _pair(state=_fixture_state())
        inp = beam_inputs()
    if name.endswith("eos_first"):
        inp["first_tokens"] = torch.tensor([5, 4, 5, 4, 5, 5])
    dec = BeamDecoder(ours, beam_width=case["beam"], kv_cache=kv_cache)
    toks = dec(pad_idx=0, max_len=2, **inp)
    padded = dec(pad_idx=0, max_len=2, unpad_output=False, **inp)
    check_beam_step1(case, toks, padded, None, inp["first_tokens"])


@pytest.mark.parametrize("layers,B,beam,Tk,d", [(2, 7, 5, 40, 512), (6, 64, 5, 128, 512), (2, 3, 1, 9, 512), (2, 5, 3, 21, 512), (1, 9, 4, 150, 512),
                                               (2, 4, 2, 64, 512), (1, 6, 7, 33, 512), (2, 7, 5, 40, 768), (3, 64, 5, 100, 768), (1, 3, 1, 33, 768)])
def test_one_launch_decoder_step_matches_the_per_operator_chain(cuda, monkeypatch, layers, B, beam, Tk, d):
    """csrc/decode_fused.hip (bf16, hidden size 512 or 768 -- the reference's default --embed --: ONE launch per decoding step, grid-wide barriers between its phases)
    against the launch-per-operator chain of imt_decode_step on the same weights, tokens, slot tables and caches: hidden
    states of every step and the q|k|v written into the cache.  Both compute in bf16 with fp32 accumulation; they differ in
    where a pre-LayerNorm sum is rounded, hence the tolerance.  Row counts that are not multiples of the 32-row items, a
    padded encoder mask and shuffled slot tables (beam re-ordering) are covered; imt_decode_check must report a clean run."""
    import imagetranslate_amd.seq2seq as S
    from imagetranslate_amd import _lib as L
    from imagetranslate_amd.param_store import store_of
    from imagetranslate_amd.seq_gen import _Incremental
    torch.manual_seed(5)
    tp = R.SyntheticTextProcessor(1000)
    ours = S.Seq2Seq(tp, lang_dec=False, enc_layer=1, dec_layer=layers, embed_dim=d, intermediate_dim=4 * d, num_attention_heads=d // 64)
    for p in ours.parameters():   # biases and LayerNorm parameters away from their (0, 1) initial values
        if p.dim() == 1:
            p.data.add_(0.1 * torch.randn_like(p))
    ours = ours.cuda().eval()
    ours.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(1)
    enc = torch.randn(B, Tk, d, generator=g).cuda().bfloat16().contiguous()
    mask = torch.ones(B, Tk, dtype=torch.uint8)
    for b in range(B):
        mask[b, Tk - (b % 4):] = 0
    mask = None if Tk == 33 else mask.cuda()   # captioning decodes against image regions without a mask (image_model.py:311-377)
    T, rows_max = 6, B * beam
    store = store_of(ours.decoder).ensure()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    toks = torch.randint(6, 1000, (T, rows_max), generator=g).cuda()
    types = torch.ones(rows_max, dtype=torch.long).cuda()
    # slot tables: at step t row r reads position j < t from cache row slots[t][r, j] (any row of the same sentence), position t from itself
    slot_tabs = []
    for t in range(T):
        rep = 1 if t == 0 else beam
        rows = B * rep
        tab = torch.zeros(rows_max, T, dtype=torch.int32)
        for r in range(rows):
            sent = r // rep
            for j in range(T):
                prev_rep = 1 if j == 0 else beam
                tab[r, j] = r if j >= t else sent * prev_rep + int(torch.randint(0, prev_rep, (1,), generator=g))
        slot_tabs.append(tab.cuda())
    runs = {}
    for mode in ("fp32", "0", "1"):   # fp32 per-operator chain (the yardstick), bf16 per-operator chain, bf16 one launch
        dt = torch.float32 if mode == "fp32" else torch.bfloat16
        ours.set_compute_dtype(dt)
        monkeypatch.setenv("IMT_DECODE_FUSED", "1" if mode == "1" else "0")
        inc = _Incremental(L.load(), ours.decoder, store, dt, store.params_for(dt), enc.to(dt), mask, B, beam, T, st)
        inc.cache.zero_()
        outs = []
        for t in range(T):
            rep = 1 if t == 0 else beam
            rows = B * rep
            out = torch.zeros(rows_max, d, device="cuda", dtype=dt)
            inc.step(t, rows, rep, toks[t, :rows].contiguous(), types[:rows].contiguous(), slot_tabs[t], out)
            outs.append(out[:rows].float().clone())
        inc.check()
        torch.cuda.synchronize()
        runs[mode] = (outs, inc.cache.view(dt).float().clone())
    from tests.util import rel_err
    for t in range(T):
        one, chain, truth = runs["1"][0][t], runs["0"][0][t], runs["fp32"][0][t]
        assert torch.isfinite(one).all()
        e_one, e_chain = rel_err(one, truth), rel_err(chain, truth)
        assert e_one <= max(1e-2, 1.5 * e_chain), "step %d: one launch %.2e from the fp32 chain, the bf16 chain %.2e" % (t, e_one, e_chain)
        assert_close(one, chain, 6e-2, "hidden states of step %d (%d layers, %d rows)" % (t, layers, one.shape[0]))
    e_one, e_chain = rel_err(runs["1"][1], runs["fp32"][1]), rel_err(runs["0"][1], runs["fp32"][1])
    assert e_one <= max(1e-2, 1.5 * e_chain), "self-attention cache: one launch %.2e from fp32, the bf16 chain %.2e" % (e_one, e_chain)
