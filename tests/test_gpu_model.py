"""GPU parity of the whole path behind the reference's module API against the CPU oracle, on the same seeded
inputs and the SAME weights (state_dict copied from the oracle).

fp32 compute mode: logits / log-probs / grads within 1e-4 relative, argmax token ids bit-exact (north_star).
bf16 compute mode (the benchmarked mode): looser, documented tolerances.
"""
import os

import pytest
import torch

from oracle import reference_model as R
from tests.util import assert_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _toy_batch(B=8, S=32, T=32, V=1000, seed=99, ragged=True):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(6, V, (B, S), generator=g)
    tgt = torch.randint(6, V, (B, T), generator=g)
    if ragged:
        ls = torch.randint(S // 2, S + 1, (B,), generator=g)
        lt = torch.randint(T // 2, T + 1, (B,), generator=g)
        src[torch.arange(S)[None] >= ls[:, None]] = 0
        tgt[torch.arange(T)[None] >= lt[:, None]] = 0
    return {"src_texts": src, "dst_texts": tgt, "src_pad_mask": src != 0, "dst_pad_mask": tgt != 0,
            "src_langs": torch.zeros(B, dtype=torch.long), "dst_langs": torch.ones(B, dtype=torch.long)}


def _pair(cls_name="Seq2Seq", enc=2, dec=2, d=128, ff=512, heads=4, V=1000, seed=0, **kw):
    import imagetranslate_amd.seq2seq as S
    import imagetranslate_amd.mass_seq2seq as M
    import imagetranslate_amd.image_model as I
    ours_cls = {"Seq2Seq": S.Seq2Seq, "MassSeq2Seq": M.MassSeq2Seq, "ImageMassSeq2Seq": I.ImageMassSeq2Seq,
                "ImageCaptioning": I.ImageCaptioning}[cls_name]
    ref_cls = getattr(R, cls_name)
    torch.manual_seed(seed)
    tp = R.SyntheticTextProcessor(V)
    ref = ref_cls(tp, lang_dec=False, enc_layer=enc, dec_layer=dec, embed_dim=d, intermediate_dim=ff,
                  num_attention_heads=heads, **kw).eval()
    ours = ours_cls(tp, lang_dec=False, enc_layer=enc, dec_layer=dec, embed_dim=d, intermediate_dim=ff,
                    num_attention_heads=heads, **kw)
    missing = ours.load_state_dict(ref.state_dict(), strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    assert all("layer_norm" in k or "obj_decoder" in k or "multistream" in k for k in missing.missing_keys), missing.missing_keys
    return ref, ours.cuda().eval()


def _grad_of(model, key):
    return dict(model.named_parameters())[key].grad


@pytest.mark.parametrize("depths", [(2, 2), (2, 1)])
def test_seq2seq_fp32_forward_backward_parity(cuda, depths):
    ref, ours = _pair(enc=depths[0], dec=depths[1])
    b = _toy_batch()
    lp_ref = ref(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"],
                 log_softmax=True)
    lp = ours(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"],
              log_softmax=True)
    assert lp.shape == lp_ref.shape and lp.dtype == torch.float32
    assert_close(lp, lp_ref, 1e-4, "log-probs")
    assert torch.equal(lp.argmax(-1).cpu(), lp_ref.argmax(-1)), "argmax token ids must be bit-exact"
    # raw logits (log_softmax=False)
    lg_ref = ref(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
    lg = ours(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
    assert_close(lg, lg_ref, 1e-4, "logits")
    # encoder states
    enc_ref = ref.encode(b["src_texts"], b["src_pad_mask"], b["src_langs"].unsqueeze(-1).expand(-1, 32))[0]
    enc = ours.encode(b["src_texts"], b["src_pad_mask"], b["src_langs"].unsqueeze(-1).expand(-1, 32))[0]
    valid = b["src_pad_mask"]
    assert_close(enc.cpu()[valid], enc_ref[valid], 1e-4, "encoder states")
    # loss + backward through the reference's own call sequence
    from imagetranslate_amd.loss import SmoothedNLLLoss
    targets = b["dst_texts"][:, 1:].contiguous().view(-1)[b["dst_pad_mask"][:, 1:].contiguous().view(-1)]
    loss_ref = R.SmoothedNLLLoss(ignore_index=0)(lp_ref, targets).mean()
    loss_ref.backward()
    crit = SmoothedNLLLoss(ignore_index=0)
    per_row = crit(lp, targets.cuda())
    assert per_row.shape == (targets.numel(), 1)
    loss = per_row.mean()
    assert_close(loss.view(1), loss_ref.view(1), 1e-5, "loss")
    loss.backward()
    ref_params = dict(ref.named_parameters())
    checked = 0
    for k, p in ours.named_parameters():
        if k not in ref_params or ref_params[k].grad is None:
            continue
        assert p.grad is not None, k
        if k.endswith("self.key.bias"):
            # softmax is invariant to a per-query constant: d/d(key bias) is exactly 0 in exact arithmetic, both
            # sides hold rounding noise only
            assert float(p.grad.abs().max()) < 1e-6 and float(ref_params[k].grad.abs().max()) < 1e-6
            continue
        assert_close(p.grad, ref_params[k].grad, 2e-4, "grad " + k)
        checked += 1
    assert checked > 40
    # pad row of the word table gets no gradient (nn.Embedding padding_idx)
    assert float(_grad_of(ours, "encoder.embeddings.word_embeddings.weight")[0].abs().max()) == 0.0


def test_fused_loss_equals_api_path_and_train_steps(cuda):
    from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup
    fx = torch.load(os.path.join(GOLD, "toy_seq2seq.pt"), weights_only=True)
    c = fx["config"]
    import imagetranslate_amd.seq2seq as S
    tp = R.SyntheticTextProcessor(c["vocab"])
    ours = S.Seq2Seq(tp, lang_dec=False, enc_layer=c["enc"], dec_layer=c["dec"], embed_dim=c["d"],
                     intermediate_dim=c["ff"], num_attention_heads=c["heads"])
    ours.load_state_dict(fx["state_dict"])
    ours = ours.cuda().eval()
    b = fx["batch"]
    lp = ours(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"],
              log_softmax=True)
    assert_close(lp, fx["log_probs"], 1e-4, "fixture log-probs")
    assert torch.equal(lp.argmax(-1).cpu(), fx["argmax"])
    opt = AdamInverseSqrtWithWarmup(ours.parameters(), lr=c["lr"], betas=(0.9, 0.98), warmup_updates=c["warmup"])
    losses = []
    for step in range(3):
        loss, ntok = ours.loss_fused(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"],
                                     b["dst_langs"])
        loss.backward()
        if step == 0:
            assert ntok == fx["log_probs"].shape[0]
            assert float(loss) == pytest.approx(fx["loss"], rel=1e-5)
            for k, gref in fx["grads"].items():
                assert_close(_grad_of(ours, k), gref, 2e-4, "fixture grad " + k)
        opt.step(max_grad_norm=1.0, zero_grad=True)
        losses.append(float(loss))
    assert losses == pytest.approx(fx["step_losses"], rel=2e-4)
    assert_close(dict(ours.named_parameters())["encoder.encoder.layer.0.attention.self.query.weight"],
                 fx["updated_query_weight"], 1e-4, "weight after 3 optimizer steps")
    # the reference's call sequence (external clip_grad_norm_ + step + zero_grad) gives the same update
    ours2 = S.Seq2Seq(tp, lang_dec=False, enc_layer=c["enc"], dec_layer=c["dec"], embed_dim=c["d"],
                      intermediate_dim=c["ff"], num_attention_heads=c["heads"])
    ours2.load_state_dict(fx["state_dict"])
    ours2 = ours2.cuda().eval()
    from imagetranslate_amd.loss import SmoothedNLLLoss
    opt2 = AdamInverseSqrtWithWarmup(ours2.parameters(), lr=c["lr"], betas=(0.9, 0.98), warmup_updates=c["warmup"])
    crit = SmoothedNLLLoss(ignore_index=0)
    targets = b["dst_texts"][:, 1:].contiguous().view(-1)[b["dst_pad_mask"][:, 1:].contiguous().view(-1)].cuda()
    l2 = []
    for step in range(3):
        pred = ours2(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"],
                     log_softmax=True)
        loss = crit(pred, targets).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ours2.parameters(), 1.0)
        opt2.step()
        opt2.zero_grad()
        l2.append(float(loss))
    assert l2 == pytest.approx(fx["step_losses"], rel=2e-4)


@pytest.mark.parametrize("V", [193, 997, 1002])
def test_vocabulary_size_not_a_multiple_of_eight(cuda, V):
    """Real tokenizers give arbitrary vocabulary sizes: logits / log-prob rows are then padded to 16-byte multiples
    internally (strided views outside); forward, API loss path, fused loss path and gradients against the oracle,
    fp32 and bf16, plus beam search on such a model."""
    ref, ours = _pair(V=V)
    b = _toy_batch(V=V)
    args = (b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
    lp_ref = ref(*args, log_softmax=True)
    lp = ours(*args, log_softmax=True)
    assert lp.shape == lp_ref.shape == (int(b["dst_pad_mask"][:, 1:].sum()), V)
    assert_close(lp, lp_ref, 1e-4, "log-probs")
    assert torch.equal(lp.argmax(-1).cpu(), lp_ref.argmax(-1))
    targets = b["dst_texts"][:, 1:].contiguous().view(-1)[b["dst_pad_mask"][:, 1:].contiguous().view(-1)]
    loss_ref = R.SmoothedNLLLoss(ignore_index=0)(lp_ref, targets).mean()
    loss_ref.backward()
    from imagetranslate_amd.loss import SmoothedNLLLoss
    loss = SmoothedNLLLoss(ignore_index=0)(lp, targets.cuda()).mean()
    loss.backward()
    assert float(loss) == pytest.approx(float(loss_ref), rel=1e-5)
    keys = ["output_layer.1.layer.weight", "output_layer.1.layer.bias", "decoder.decoder.layer.1.output.dense.weight",
            "encoder.embeddings.word_embeddings.weight"]
    g_api = {k: _grad_of(ours, k).clone() for k in keys}
    for k in keys:
        assert_close(g_api[k], _grad_of(ref, k), 2e-4, "API-path grad " + k)
    ours.zero_grad()
    fused, ntok = ours.loss_fused(*args)
    fused.backward()
    assert ntok == lp.shape[0] and float(fused) == pytest.approx(float(loss_ref), rel=1e-5)
    for k in keys:
        assert_close(_grad_of(ours, k), _grad_of(ref, k), 2e-4, "fused-path grad " + k)
    ours.zero_grad()
    ours.set_compute_dtype(torch.bfloat16)
    fused16, _ = ours.loss_fused(*args)
    fused16.backward()
    assert float(fused16) == pytest.approx(float(loss_ref), rel=2e-2)
    assert_close(_grad_of(ours, keys[0]), _grad_of(ref, keys[0]), 6e-2, "bf16 fused grad")
    # beam search with the padded logits rows
    from imagetranslate_amd.seq_gen import BeamDecoder
    from oracle.seq_gen import BeamDecoder as OracleBeam
    ours.set_compute_dtype(torch.float32)
    n = 4
    kw = dict(src_inputs=b["src_texts"][:n], src_sizes=b["src_pad_mask"][:n].sum(1), first_tokens=torch.full((n,), 6),
              src_mask=b["src_pad_mask"][:n], src_langs=b["src_langs"][:n], tgt_langs=b["dst_langs"][:n], pad_idx=0, max_len=9)
    exp = OracleBeam(ref, beam_width=3)(**kw)
    got = BeamDecoder(ours.eval(), beam_width=3)(**kw)
    assert [g.tolist() for g in got] == [e.tolist() for e in exp]


@pytest.mark.parametrize("variant", [dict(lang_dec=True), dict(lang_dec=True, tie_embed=True), dict(lang_dec=False, tie_embed=True)])
def test_topology_variants_forward_backward_and_beam(cuda, variant):
    """Per-language decoders (deep copies, src/seq2seq.py:67-77) and the --tie quirk (:55-59): log-probs, loss and
    gradients against the oracle (fp32), training through the fused path, and beam search on the chosen decoder."""
    import imagetranslate_amd.seq2seq as S
    from imagetranslate_amd.seq_gen import BeamDecoder
    from oracle.seq_gen import BeamDecoder as OracleBeam
    torch.manual_seed(3)
    tp = R.SyntheticTextProcessor(1000)
    kw = dict(enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512, num_attention_heads=4, **variant)
    ref = R.Seq2Seq(tp, **kw).eval()
    ours = S.Seq2Seq(tp, **kw)
    missing = ours.load_state_dict(ref.state_dict(), strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    ours = ours.cuda().eval()
    b = _toy_batch()
    args = (b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
    lp_ref, lp = ref(*args, log_softmax=True), ours(*args, log_softmax=True)
    assert_close(lp, lp_ref, 1e-4, "log-probs")
    assert torch.equal(lp.argmax(-1).cpu(), lp_ref.argmax(-1))
    targets = b["dst_texts"][:, 1:].contiguous().view(-1)[b["dst_pad_mask"][:, 1:].contiguous().view(-1)]
    loss_ref = R.SmoothedNLLLoss(ignore_index=0)(lp_ref, targets).mean()
    loss_ref.backward()
    loss, ntok = ours.loss_fused(*args)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(loss_ref.detach()), rel=1e-5) and ntok == lp.shape[0]
    checked = 0
    ref_named = dict(ref.named_parameters())
    for k, p in ours.named_parameters():
        g_ref = ref_named[k].grad
        if g_ref is None or float(g_ref.abs().max()) < 1e-9:  # the other language's copies; key biases (softmax shift invariance)
            assert p.grad is None or float(p.grad.abs().max()) < 1e-6, "unexpected gradient on " + k
            continue
        assert_close(p.grad, g_ref, 3e-4, "grad " + k)
        checked += 1
    assert checked > 40
    n = 4
    bk = dict(src_inputs=b["src_texts"][:n], src_sizes=b["src_pad_mask"][:n].sum(1), first_tokens=torch.full((n,), 6),
              src_mask=b["src_pad_mask"][:n], src_langs=b["src_langs"][:n], tgt_langs=b["dst_langs"][:n], pad_idx=0, max_len=8)
    exp = OracleBeam(ref, beam_width=3)(**bk)
    got = BeamDecoder(ours, beam_width=3)(**bk)
    assert [g.tolist() for g in got] == [e.tolist() for e in exp]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_overlapped_optimizer_step_matches_in_order_step(cuda, dtype):
    """optimizer.step(overlap_next_forward=True) -- segments on a side stream, each stack of the next forward waiting
    only for what it reads -- must leave the parameters of the in-order step, step after step.  (Not bit-for-bit: the
    embedding / LayerNorm gradients are atomic sums whose order varies run to run; a stale-parameter race would show
    at the scale of the updates, 1e-3, not 1e-7.)"""
    from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup
    results = []
    for overlap in (False, True):
        _, ours = _pair()
        ours.set_compute_dtype(dtype)
        ours.train()
        for mod in (ours.encoder, ours.decoder):
            mod._imt_dropout_seed = 123  # same dropout masks in both runs
        opt = AdamInverseSqrtWithWarmup(ours.parameters(), lr=1e-3, betas=(0.9, 0.98), warmup_updates=2)
        losses = []
        for step in range(4):
            b = _toy_batch(seed=50 + step)
            loss, _ = ours.loss_fused(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
            loss.backward()
            opt.step(max_grad_norm=1.0, zero_grad=True, overlap_next_forward=overlap)
            losses.append(float(loss.detach()))
        results.append((losses, {k: v.detach().clone() for k, v in ours.state_dict().items()}))
    assert results[0][0] == pytest.approx(results[1][0], rel=1e-5), "losses differ: %s vs %s" % (results[0][0], results[1][0])
    for k, v in results[0][1].items():
        diff = float((v.float() - results[1][1][k].float()).abs().max())
        assert diff <= 2e-5 * max(1.0, float(v.float().abs().max())), (k, diff)  # atomics-order noise is ~3e-6 after 4 steps


def test_train_steps_retain_no_memory_without_the_cyclic_collector(cuda):
    """With gc disabled, device memory held between steps must not grow: an autograd node that keeps its own output
    as a plain attribute forms a cycle only the cyclic collector frees (a B*T*d leak per step until it runs)."""
    import gc
    from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup
    _, ours = _pair()
    ours.set_compute_dtype(torch.bfloat16)
    ours.train()
    opt = AdamInverseSqrtWithWarmup(ours.parameters(), lr=1e-3, betas=(0.9, 0.98), warmup_updates=2)
    b = _toy_batch(seed=7)
    held = []
    gc.collect()
    gc.disable()
    try:
        for step in range(12):
            loss, _ = ours.loss_fused(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
            loss.backward()
            opt.step(max_grad_norm=1.0, zero_grad=True, overlap_next_forward=True)
            del loss
            torch.cuda.synchronize()
            held.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert held[-1] == held[3], held


@pytest.mark.parametrize("B,S,T", [(2, 512, 512), (1, 1, 2), (3, 7, 2), (2, 130, 129)])
def test_extreme_lengths_fp32_parity(cuda, B, S, T):
    """Maximum length (max_position_embeddings = 512: multi-tile attention, two-kernel backward), the shortest legal
    batch (one source token, one target transition), and lengths just past a tile boundary; ragged padding throughout."""
    ref, ours = _pair()
    g = torch.Generator().manual_seed(S * 1000 + T)
    src = torch.randint(6, 1000, (B, S), generator=g)
    tgt = torch.randint(6, 1000, (B, T), generator=g)
    if S > 4:
        src[0, S - S // 3:] = 0
    if T > 4:
        tgt[-1, T - T // 4:] = 0
    args = (src, tgt, src != 0, tgt != 0, torch.zeros(B, dtype=torch.long), torch.ones(B, dtype=torch.long))
    lp_ref, lp = ref(*args, log_softmax=True), ours(*args, log_softmax=True)
    assert lp.shape == lp_ref.shape
    assert_close(lp, lp_ref, 1e-4, "log-probs")
    assert torch.equal(lp.argmax(-1).cpu(), lp_ref.argmax(-1))
    targets = tgt[:, 1:][(tgt != 0)[:, 1:]]
    loss_ref = R.SmoothedNLLLoss(ignore_index=0)(lp_ref, targets).mean()
    loss_ref.backward()
    loss, ntok = ours.loss_fused(*args)
    loss.backward()
    assert ntok == targets.numel() and float(loss.detach()) == pytest.approx(float(loss_ref.detach()), rel=2e-5)
    for k in ["encoder.embeddings.position_embeddings.weight", "encoder.encoder.layer.1.attention.self.value.weight",
              "decoder.decoder.layer.0.crossattention.self.query.weight", "output_layer.1.layer.bias"]:
        g_ref = _grad_of(ref, k)
        if float(g_ref.abs().max()) < 1e-9:  # exactly zero in theory (softmax over a single key has no query gradient)
            assert float(_grad_of(ours, k).abs().max()) < 1e-6, k
        else:
            assert_close(_grad_of(ours, k), g_ref, 3e-4, "grad " + k)


def test_bf16_mode_tracks_fp32(cuda):
    ref, ours = _pair()
    ours.set_compute_dtype(torch.bfloat16)
    b = _toy_batch()
    lp_ref = ref(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"],
                 log_softmax=True)
    lp = ours(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"],
              log_softmax=True)
    assert lp.dtype == torch.float32
    assert_close(lp, lp_ref, 3e-2, "bf16 log-probs")
    targets = b["dst_texts"][:, 1:].contiguous().view(-1)[b["dst_pad_mask"][:, 1:].contiguous().view(-1)]
    loss_ref = R.SmoothedNLLLoss(ignore_index=0)(lp_ref, targets).mean()
    loss_ref.backward()
    loss, _ = ours.loss_fused(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"],
                              b["dst_langs"])
    assert float(loss) == pytest.approx(float(loss_ref), rel=2e-2)
    loss.backward()
    ref_params = dict(ref.named_parameters())
    for k in ["encoder.encoder.layer.0.attention.self.query.weight", "decoder.decoder.layer.1.crossattention.self.key.weight",
              "output_layer.1.layer.weight", "encoder.encoder.layer.1.output.dense.weight",
              "encoder.embeddings.word_embeddings.weight"]:
        assert_close(_grad_of(ours, k), ref_params[k].grad, 8e-2, "bf16 grad " + k)


def test_mass_forward_with_positions(cuda):
    ref, ours = _pair("MassSeq2Seq")
    g = torch.Generator().manual_seed(5)
    B, S = 6, 40
    src = torch.randint(6, 1000, (B, S), generator=g)
    lens = torch.randint(20, S + 1, (B,), generator=g)
    src[torch.arange(S)[None] >= lens[:, None]] = 0
    import random
    from imagetranslate_amd.utils import mass_mask
    random.seed(11)
    info = mass_mask(0.5, lens, src.clone(), R.SyntheticTextProcessor(1000))
    langs = torch.zeros(B, dtype=torch.long)
    out_ref = ref(src_inputs=info["src_text"], tgt_inputs=info["to_recover"], tgt_positions=info["positions"],
                  src_langs=langs, log_softmax=True)
    out = ours(src_inputs=info["src_text"], tgt_inputs=info["to_recover"], tgt_positions=info["positions"], src_langs=langs,
               log_softmax=True)
    assert out.shape[0] == info["targets"].numel()
    assert_close(out, out_ref, 1e-4, "MASS log-probs")
    assert torch.equal(out.argmax(-1).cpu(), out_ref.argmax(-1))
    # MT route through the same class (tgt_langs given)
    b = _toy_batch()
    o_ref = ref(src_inputs=b["src_texts"], tgt_inputs=b["dst_texts"], src_langs=b["src_langs"], tgt_langs=b["dst_langs"],
                log_softmax=True)
    o = ours(src_inputs=b["src_texts"], tgt_inputs=b["dst_texts"], src_langs=b["src_langs"], tgt_langs=b["dst_langs"],
             log_softmax=True)
    assert_close(o, o_ref, 1e-4, "MASS-class MT log-probs")


def test_image_captioning_path(cuda):
    ref, ours = _pair("ImageCaptioning", image_feat_dim=256, use_obj=False)
    g = torch.Generator().manual_seed(6)
    B, T = 4, 12
    feats = torch.randn(B, 49, 256, generator=g)
    tgt = torch.randint(6, 1000, (B, T), generator=g)
    lt = torch.randint(6, T + 1, (B,), generator=g)
    tgt[torch.arange(T)[None] >= lt[:, None]] = 0
    langs = torch.ones(B, dtype=torch.long)
    kw = dict(tgt_inputs=tgt, tgt_mask=tgt != 0, tgt_langs=langs, batch={"images": feats}, log_softmax=True)
    out_ref = ref(**kw)
    out = ours(**kw)
    assert_close(out, out_ref, 1e-4, "captioning log-probs")
    emb = ours(batch={"images": feats}, encode_only=True)
    assert emb.shape == (B, 49, 128)
    # text route of the captioning class
    b = _toy_batch()
    o_ref = ref(src_inputs=b["src_texts"], tgt_inputs=b["dst_texts"], src_langs=b["src_langs"], tgt_langs=b["dst_langs"],
                log_softmax=True)
    o = ours(src_inputs=b["src_texts"], tgt_inputs=b["dst_texts"], src_langs=b["src_langs"], tgt_langs=b["dst_langs"],
             log_softmax=True)
    assert_close(o, o_ref, 1e-4, "captioning-class MT log-probs")
    # gradients reach the image head
    loss, ntok = ours.loss_fused(tgt_inputs=tgt, tgt_mask=tgt != 0, tgt_langs=langs, batch={"images": feats})
    loss.backward()
    tref = tgt[:, 1:][(tgt != 0)[:, 1:]]
    lref = R.SmoothedNLLLoss(ignore_index=0)(out_ref, tref).mean()
    lref.backward()
    assert float(loss) == pytest.approx(float(lref), rel=1e-4)
    assert_close(ours.image_model.fc.weight.grad, ref.image_model.fc.weight.grad, 2e-4, "fc grad")
    assert_close(ours.image_model.location_embedding.weight.grad, ref.image_model.location_embedding.weight.grad, 2e-4,
                 "location embedding grad")


def test_decoder_mask_variants_and_standalone_calls(cuda):
    """BertDecoderModel called the way BeamDecoder does (2-D ones mask => causal, src/seq_gen.py:164-166) and with an
    explicit 3-D future_mask: both equal the oracle."""
    from imagetranslate_amd.seq2seq import future_mask
    ref, ours = _pair()
    b = _toy_batch()
    S = b["src_texts"].shape[1]
    enc_ref = ref.encode(b["src_texts"], b["src_pad_mask"], b["src_langs"].unsqueeze(-1).expand(-1, S))[0]
    with torch.no_grad():
        enc = ours.encode(b["src_texts"], b["src_pad_mask"], b["src_langs"].unsqueeze(-1).expand(-1, S))[0]
        tgt = b["dst_texts"][:, :10]
        ones = torch.ones_like(tgt)
        tl = b["dst_langs"].unsqueeze(-1).expand(-1, 10)
        d_ref = ref.decoder(encoder_states=enc_ref, input_ids=tgt, encoder_attention_mask=b["src_pad_mask"],
                            tgt_attention_mask=ones, token_type_ids=tl)
        d = ours.decoder(encoder_states=enc, input_ids=tgt.cuda(), encoder_attention_mask=b["src_pad_mask"].cuda(),
                         tgt_attention_mask=ones.cuda(), token_type_ids=tl.cuda())
        assert_close(d, d_ref, 1e-4, "decoder (2-D ones mask)")
        fm = future_mask(b["dst_pad_mask"][:, :10])
        valid = b["dst_pad_mask"][:, :10]
        d_ref = ref.decoder(encoder_states=enc_ref, input_ids=tgt, encoder_attention_mask=b["src_pad_mask"],
                            tgt_attention_mask=fm, token_type_ids=tl)
        d = ours.decoder(encoder_states=enc, input_ids=tgt.cuda(), encoder_attention_mask=b["src_pad_mask"].cuda(),
                         tgt_attention_mask=fm.cuda(), token_type_ids=tl.cuda())
        assert_close(d.cpu()[valid], d_ref[valid], 1e-4, "decoder (3-D future_mask)")
        out = ours.output_layer[1](d[:, -1, :])
        assert out.shape == (8, 1000)


def test_stack_operand_shapes_are_checked_on_the_host(cuda):
    """The runtime takes raw pointers, so a mis-shaped operand must be refused before anything is launched (a [batch]
    token_type_ids read as [batch, length] faulted the GPU once); per-sentence language ids, [1, length] positions and
    int32 ids are accepted the way torch / HF broadcasting would take them and give the same states."""
    _, ours = _pair()
    b = _toy_batch()
    S = b["src_texts"].shape[1]
    ids, mask, langs = b["src_texts"].cuda(), b["src_pad_mask"].cuda(), b["src_langs"].cuda()
    with torch.no_grad():
        want = ours.encode(ids, mask, langs.unsqueeze(-1).expand(-1, S))[0]
        assert torch.equal(ours.encode(ids, mask, langs)[0], want)                          # [batch] language ids
        assert torch.equal(ours.encode(ids.int(), mask, langs.int().unsqueeze(-1))[0], want)  # int32, [batch, 1]
        pos = torch.arange(S, device="cuda").unsqueeze(0)
        assert torch.equal(ours.encoder(ids, attention_mask=mask, token_type_ids=langs, position_ids=pos), want)
        for bad in (dict(attention_mask=mask[:, :-1]), dict(token_type_ids=langs[:-1]), dict(position_ids=pos[:, :-1]),
                    dict(attention_mask=mask[:-1])):
            kw = dict(attention_mask=mask, token_type_ids=langs)
            kw.update(bad)
            with pytest.raises(ValueError):
                ours.encoder(ids, **kw)
        with pytest.raises(ValueError):
            ours.decoder(encoder_states=want[:-1], input_ids=ids, encoder_attention_mask=mask, token_type_ids=langs)
        with pytest.raises(ValueError):
            ours.decoder(encoder_states=want, input_ids=ids, encoder_attention_mask=mask[:, :-1], token_type_ids=langs)
        with pytest.raises(ValueError):
            ours.decoder(encoder_states=want[:, :, :-8], input_ids=ids, encoder_attention_mask=mask, token_type_ids=langs)


def test_save_load_roundtrip(cuda, tmp_path):
    import imagetranslate_amd.seq2seq as S
    ref, ours = _pair()
    ours.save(str(tmp_path))
    loaded = S.Seq2Seq.load(S.Seq2Seq, str(tmp_path), tok_dir=None, text_processor=R.SyntheticTextProcessor(1000),
                            num_attention_heads=4)
    b = _toy_batch()
    a = ours(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
    c = loaded.eval()(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
    assert torch.equal(a, c)


def test_training_mode_dropout_is_consistent_between_forward_and_backward(cuda):
    """Dropout masks are regenerated in backward from (seed, element index).  With the seed pinned the train-mode loss
    is a deterministic function of the parameters, so analytic gradients must match central finite differences --
    this fails if ANY of the ~10 dropout sites per layer (embedding, attention probabilities, dense outputs) uses a
    different mask in backward than in forward."""
    ref, ours = _pair(enc=1, dec=1, d=64, ff=128, heads=2)
    ours.train()
    for st in ours._stacks():
        st._imt_dropout_seed = 12345
    b = _toy_batch(B=4, S=16, T=16)
    args = (b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])

    def loss_value():
        with torch.no_grad():
            return float(ours.loss_fused(*args)[0])

    l1, l2 = loss_value(), loss_value()
    assert l1 == l2, "train-mode forward is not deterministic for a pinned seed"
    ours.eval()
    l_eval = loss_value()
    ours.train()
    assert abs(l1 - l_eval) > 1e-6, "dropout had no effect in training mode"
    ours.zero_grad()
    loss, _ = ours.loss_fused(*args)
    loss.backward()
    named = dict(ours.named_parameters())
    checks = [("encoder.encoder.layer.0.attention.self.value.weight", (3, 5)),
              ("encoder.encoder.layer.0.intermediate.dense.weight", (7, 11)),
              ("decoder.decoder.layer.0.crossattention.self.query.weight", (2, 9)),
              ("decoder.decoder.layer.0.output.dense.bias", (13,)),
              ("encoder.embeddings.LayerNorm.weight", (6,)),
              ("output_layer.1.layer.bias", (int(b["dst_texts"][0, 1]),))]
    eps = 2e-2
    for key, idx in checks:
        p = named[key]
        g = float(p.grad[idx])
        with torch.no_grad():
            old = float(p[idx])
            p[idx] = old + eps
        lp = loss_value()
        with torch.no_grad():
            p[idx] = old - eps
        lm = loss_value()
        with torch.no_grad():
            p[idx] = old
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g) <= 0.08 * max(abs(fd), abs(g)) + 2e-5, "%s%s: analytic %.6g vs finite-difference %.6g" % (key, idx, g, fd)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_intermediate_size_not_a_multiple_of_the_k_tile(cuda, dtype):
    """ff = 1032: the FFN-down projection has K = 1032 (not a whole number of 64- / 32-element K tiles, and >= 16 tiles long),
    so imt_gemm takes its ragged-tail + whole-tile-body split; with dropout off (eval, dev loss, decoding) the dense +
    LayerNorm call carries ln_out through that split.  Eval forward, train-mode step and beam search against the oracle."""
    ref, ours = _pair(ff=1032)
    ours.set_compute_dtype(dtype)
    tol, gtol = (1e-4, 2e-4) if dtype == torch.float32 else (3e-2, 8e-2)
    b = _toy_batch()
    args = (b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
    lp_ref = ref(*args, log_softmax=True)
    with torch.no_grad():
        lp = ours(*args, log_softmax=True)       # eval mode: dropout 0 -> the split path with ln_out
    assert_close(lp, lp_ref, tol, "log-probs, ff=1032")
    targets = b["dst_texts"][:, 1:][b["dst_pad_mask"][:, 1:]]
    loss_ref = R.SmoothedNLLLoss(ignore_index=0)(lp_ref, targets).mean()
    loss_ref.backward()
    ours.zero_grad()
    loss, _ = ours.loss_fused(*args)
    loss.backward()
    assert float(loss) == pytest.approx(float(loss_ref), rel=10 * tol)
    for k in ("decoder.decoder.layer.1.output.dense.weight", "encoder.encoder.layer.0.intermediate.dense.weight",
              "encoder.encoder.layer.1.output.LayerNorm.weight"):
        assert_close(_grad_of(ours, k), _grad_of(ref, k), gtol, "grad " + k)
    if dtype == torch.float32:
        from imagetranslate_amd.seq_gen import BeamDecoder
        from oracle.seq_gen import BeamDecoder as OracleBeam
        n = 4
        kw = dict(src_inputs=b["src_texts"][:n], src_sizes=b["src_pad_mask"][:n].sum(1), first_tokens=torch.full((n,), 6),
                  src_mask=b["src_pad_mask"][:n], src_langs=b["src_langs"][:n], tgt_langs=b["dst_langs"][:n], pad_idx=0, max_len=9)
        exp = OracleBeam(ref, beam_width=3)(**kw)
        got = BeamDecoder(ours, beam_width=3)(**kw)          # imt_decode_step through the same dense + LayerNorm call
        assert [g.tolist() for g in got] == [e.tolist() for e in exp]
    ours.set_compute_dtype(torch.float32)
