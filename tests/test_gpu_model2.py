"""Model-level BACKWARD parity beyond plain MT (toy size, fp32 compute mode, against the CPU oracle on the same weights):
MASS (decoder fed the masked span with its original positions), captioning (decoder + image head), lexical proposals
(``use_proposals=True``, src/seq2seq.py:110-144); gradient accumulation with the reference's clip-every-micro-step
order (src/train_image_mt.py:291-295); the image-head kernels."""
import random

import pytest
import torch

from oracle import reference_model as R
from tests.test_gpu_model import _pair, _toy_batch
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _compare_all_grads(ours, ref, tol, min_checked):
    ref_params = dict(ref.named_parameters())
    checked = 0
    for k, p in ours.named_parameters():
        if k not in ref_params or ref_params[k].grad is None:
            continue
        g_ref = ref_params[k].grad
        assert p.grad is not None, k
        if float(g_ref.abs().max()) < 1e-7:  # exactly zero in exact arithmetic (key biases; unused rows): noise on both sides
            assert float(p.grad.abs().max()) < 1e-5, k
            continue
        assert_close(p.grad, g_ref, tol, "grad " + k)
        checked += 1
    assert checked >= min_checked, checked


def test_mass_backward_parity(cuda):
    from imagetranslate_amd.utils import mass_mask
    ref, ours = _pair("MassSeq2Seq")
    g = torch.Generator().manual_seed(5)
    B, S = 6, 40
    src = torch.randint(6, 1000, (B, S), generator=g)
    lens = torch.randint(20, S + 1, (B,), generator=g)
    src[torch.arange(S)[None] >= lens[:, None]] = 0
    random.seed(11)
    info = mass_mask(0.5, lens, src.clone(), R.SyntheticTextProcessor(1000))
    langs = torch.zeros(B, dtype=torch.long)
    lp_ref = ref(src_inputs=info["src_text"], tgt_inputs=info["to_recover"], tgt_positions=info["positions"], src_langs=langs,
                 log_softmax=True)
    loss_ref = R.SmoothedNLLLoss(ignore_index=0)(lp_ref, info["targets"]).mean()
    loss_ref.backward()
    ours.zero_grad()
    loss, ntok = ours.loss_fused(src_inputs=info["src_text"], tgt_inputs=info["to_recover"], src_langs=langs,
                                 tgt_positions=info["positions"])
    loss.backward()
    assert ntok == info["targets"].numel()
    assert float(loss) == pytest.approx(float(loss_ref), rel=1e-5)
    _compare_all_grads(ours, ref, 2e-4, 40)
    # positions outside arange: the position table's gradient rows are exactly those the span positions address
    gp = dict(ours.named_parameters())["encoder.embeddings.position_embeddings.weight"].grad
    assert float(gp[S:].abs().max()) == 0.0


def test_captioning_backward_parity_decoder_and_head(cuda):
    ref, ours = _pair("ImageCaptioning", image_feat_dim=256, use_obj=False)
    g = torch.Generator().manual_seed(6)
    B, T = 5, 14
    feats = torch.randn(B, 49, 256, generator=g)
    tgt = torch.randint(6, 1000, (B, T), generator=g)
    lt = torch.randint(6, T + 1, (B,), generator=g)
    tgt[torch.arange(T)[None] >= lt[:, None]] = 0
    langs = torch.ones(B, dtype=torch.long)
    out_ref = ref(tgt_inputs=tgt, tgt_mask=tgt != 0, tgt_langs=langs, batch={"images": feats}, log_softmax=True)
    lref = R.SmoothedNLLLoss(ignore_index=0)(out_ref, tgt[:, 1:][(tgt != 0)[:, 1:]]).mean()
    lref.backward()
    ours.zero_grad()
    loss, _ = ours.loss_fused(tgt_inputs=tgt, tgt_mask=tgt != 0, tgt_langs=langs, batch={"images": feats})
    loss.backward()
    assert float(loss) == pytest.approx(float(lref), rel=1e-5)
    # every decoder parameter, the embeddings, the output layer and the image head (the text encoder is unused: no grads)
    _compare_all_grads(ours, ref, 2e-4, 30)
    for k in ("image_model.fc.weight", "image_model.location_embedding.weight", "decoder.decoder.layer.1.crossattention.self.key.weight",
              "decoder.decoder.layer.0.crossattention.self.value.bias", "decoder.decoder.layer.1.output.dense.weight"):
        assert dict(ref.named_parameters())[k].grad is not None and float(dict(ours.named_parameters())[k].grad.abs().max()) > 0, k


def test_image_head_dropout_kernels(cuda):
    """imt_add_rows_dropout: plain dropout keeps ~1-p of the elements scaled by 1/(1-p), is reproducible from its seed, and
    the train-mode head's backward uses the forward's masks (d fc.weight == dy^T x with the masked operands)."""
    from imagetranslate_amd import hip_ops as O
    x = torch.randn(300, 512, device="cuda")
    y = O.add_rows_dropout(x, None, dropout_p=0.25, dropout_seed=7)
    kept = (y != 0)
    assert abs(float(kept.float().mean()) - 0.75) < 0.01
    assert torch.allclose(y[kept], (x / 0.75)[kept])
    assert torch.equal(y, O.add_rows_dropout(x, None, dropout_p=0.25, dropout_seed=7))
    assert not torch.equal(y, O.add_rows_dropout(x, None, dropout_p=0.25, dropout_seed=8))
    loc = torch.randn(49, 512, device="cuda")
    z = O.add_rows_dropout(x[:98], loc, dropout_p=0.0)
    assert torch.allclose(z, x[:98] + loc.repeat(2, 1))
    zb = O.add_rows_dropout(x[:98], loc.bfloat16(), out_dtype=torch.bfloat16, dropout_p=0.0)
    assert zb.dtype == torch.bfloat16 and torch.allclose(zb.float(), (x[:98] + loc.bfloat16().float().repeat(2, 1)), atol=0.03, rtol=0.01)
    # the head in train mode: gradients are those of the SAME masks
    from imagetranslate_amd.image_model import ImageCaptioning
    torch.manual_seed(3)
    m = ImageCaptioning(R.SyntheticTextProcessor(200), lang_dec=False, enc_layer=1, dec_layer=1, embed_dim=128, intermediate_dim=256,
                        num_attention_heads=4, image_feat_dim=64, use_obj=False).cuda().train()
    m.image_model._imt_dropout_seed = 99
    feats = torch.randn(3, 49, 64)
    out, _ = m.image_model(feats)
    out2, _ = m.image_model(feats)
    assert torch.equal(out, out2) and float((out == 0).float().mean()) > 0.05
    m.zero_grad()
    w = torch.randn_like(out)
    (out * w).sum().backward()
    # manual: dy = dropout_bwd(w) ; d fc = dy^T xd ; d loc = sum_b dy
    p = m.image_model.dropout
    xd = O.add_rows_dropout(feats.cuda().reshape(-1, 64), None, dropout_p=p, dropout_seed=99)
    dy = O.add_rows_dropout(w.reshape(-1, 128).contiguous(), None, dropout_p=p, dropout_seed=100)
    assert_close(m.image_model.fc.weight.grad, dy.t() @ xd, 1e-4, "fc grad (train mode)")
    assert_close(m.image_model.location_embedding.weight.grad, dy.view(3, 49, 128).sum(0), 1e-4, "location grad (train mode)")


def test_use_proposals_forward_backward_parity(cuda):
    """Lexical proposals (torch ops on the GPU around the HIP decoder): same numbers as the oracle's restatement of
    src/seq2seq.py:110-144, forward and backward, including the shared word-embedding table that receives gradients from the
    encoder / decoder embedding kernels AND from the proposal lookup."""
    ref, ours = _pair(use_proposals=True)
    b = _toy_batch()
    g = torch.Generator().manual_seed(8)
    B = b["src_texts"].shape[0]
    props = torch.randint(6, 1000, (B, 7), generator=g)
    props[1, 4:] = 0
    props[2, :] = 0  # a row without proposals: the 1e-8 constant path
    args = (b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
    lp_ref = ref(*args, proposals=props, log_softmax=True)
    lp = ours(*args, proposals=props, log_softmax=True)
    assert_close(lp, lp_ref, 1e-4, "log-probs with proposals")
    assert torch.equal(lp.argmax(-1).cpu(), lp_ref.argmax(-1))
    targets = b["dst_texts"][:, 1:][b["dst_pad_mask"][:, 1:]]
    loss_ref = R.SmoothedNLLLoss(ignore_index=0)(lp_ref, targets).mean()
    loss_ref.backward()
    ours.zero_grad()
    loss, _ = ours.loss_fused(*args, proposals=props)
    loss.backward()
    assert float(loss) == pytest.approx(float(loss_ref), rel=1e-5)
    _compare_all_grads(ours, ref, 3e-4, 42)
    assert float(dict(ours.named_parameters())["lexical_gate"].grad.abs().max()) > 0


def test_gradient_accumulation_matches_reference_order(cuda):
    """--acc 2, two windows: backward, clip the ACCUMULATED gradient, backward again, clip again, step
    (src/train_image_mt.py:283-295) against the oracle doing exactly that with torch's clip_grad_norm_ + its Adam
    restatement.  (The first optimizer step runs at the warm-up's initial lr 1e-7, the second at 2.5e-4: the parameters are
    compared after the second, where the update is ~3e-3 of the weights' scale and the 1e-4 tolerance ~3 % of the update.)"""
    from imagetranslate_amd.parallel import train_step
    from imagetranslate_amd.utils import build_optimizer
    ref, ours = _pair()
    init = {k: v.clone() for k, v in ref.state_dict().items()}
    clip = 0.05  # small enough that every clip bites
    opt_ref = R.AdamInverseSqrtWithWarmup(ref.parameters(), lr=1e-3, betas=(0.9, 0.98), warmup_updates=4)
    crit = R.SmoothedNLLLoss(ignore_index=0)
    batches = [_toy_batch(seed=s) for s in (1, 2, 3, 4)]
    for i, b in enumerate(batches):
        lp = ref(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"], log_softmax=True)
        crit(lp, b["dst_texts"][:, 1:][b["dst_pad_mask"][:, 1:]]).mean().backward()
        assert float(torch.nn.utils.clip_grad_norm_(ref.parameters(), clip)) > clip
        if i % 2 == 1:
            opt_ref.step()
            opt_ref.zero_grad()
    opt = build_optimizer(ours, 1e-3, 4)
    ours.zero_grad()
    to_dev = lambda b: {k: v.cuda() if k not in ("src_langs", "dst_langs") else v for k, v in b.items()}
    for i, b in enumerate(batches):
        train_step(ours, opt, to_dev(b), clip=clip, update=(i % 2 == 1))
        if i == 0:
            g_mid = dict(ours.named_parameters())["output_layer.1.layer.weight"].grad
            assert 0 < float(g_mid.norm()) < clip * 1.01  # clipped in place, kept for the next micro-step
    torch.cuda.synchronize()
    ref_sd, our_sd = ref.state_dict(), ours.state_dict()
    for k in ("encoder.encoder.layer.0.attention.self.query.weight", "decoder.decoder.layer.1.output.dense.weight",
              "output_layer.1.layer.weight", "encoder.embeddings.word_embeddings.weight"):
        moved = float((ref_sd[k] - init[k]).abs().max()) / float(init[k].abs().max())
        assert moved > 1e-3, (k, moved)
        assert_close(our_sd[k], ref_sd[k], 1e-4, "parameters after two accumulation windows: " + k)


def test_overlapped_optimizer_step_equals_in_order(cuda, monkeypatch):
    """The optimizer step on a side stream -- per-segment events (embeddings, encoder layer by layer, decoder, output layers),
    the encoder forward waiting site by site inside imt_stack_forward (imt_stack_io.wait_events) -- is the same arithmetic as
    the in-order step.  Three steps on different batches in fp32 compute mode: parameters equal to 1e-4 of their scale (the
    embedding / LayerNorm gradients are fp32 atomic sums whose order varies from run to run, so not bit for bit); a forward
    that read a parameter before its update had landed would be off by ~lr / scale = 2.5e-2."""
    from imagetranslate_amd.parallel import train_step
    from imagetranslate_amd.utils import build_optimizer
    results = {}
    for name, overlap in (("overlap", "1"), ("in_order", "0")):
        monkeypatch.setenv("IMT_ADAM_OVERLAP", overlap)
        _, ours = _pair(seed=3)
        ours.eval()  # no dropout: the runs see the same arithmetic
        opt = build_optimizer(ours, 2e-3, 2)
        for s in (11, 12, 13):
            b = _toy_batch(seed=s)
            b = {k: v.cuda() if k not in ("src_langs", "dst_langs") else v for k, v in b.items()}
            train_step(ours, opt, b, clip=0.5)
        torch.cuda.synchronize()
        results[name] = {k: v.clone() for k, v in ours.state_dict().items()}
        moved = float((results[name]["output_layer.1.layer.weight"] - _pair(seed=3)[0].state_dict()["output_layer.1.layer.weight"].cuda()).abs().max())
        assert moved > 1e-3  # the steps did move the weights by ~lr each
    for k in results["overlap"]:
        if k.endswith("self.key.bias"):
            continue  # exactly-zero gradient in exact arithmetic (softmax invariance): Adam normalises pure rounding noise
        assert_close(results["overlap"][k], results["in_order"][k], 1e-4, "overlapped vs in-order step: " + k)


def test_partial_gradient_norm_under_the_backward(cuda):
    """From the second step on, the squared norm of everything below the encoder's range (output layers, decoder) is summed on a
    side stream as soon as the decoder's backward is enqueued, and step() adds the rest: the total equals the norm of the whole
    gradient buffer; a zero_grad between backward and step voids the partial sum."""
    from imagetranslate_amd.param_store import store_of
    from imagetranslate_amd.utils import build_optimizer
    _, ours = _pair(seed=4)
    ours.eval()
    opt = build_optimizer(ours, 1e-3, 2)
    st = store_of(ours.encoder).ensure()
    for i, s in enumerate((21, 22, 23, 24)):
        b = _toy_batch(seed=s)
        args = (b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
        loss, _ = ours.loss_fused(*args)
        loss.backward()
        if i == 0:
            assert opt._partial is None  # the hook is installed by the first step
        else:
            assert opt._partial is not None and 0 < opt._partial[0] < st.total
        if i == 2:  # gradients thrown away and recomputed: the partial sum of the first backward must not be used
            ours.zero_grad()
            loss, _ = ours.loss_fused(*args)
            loss.backward()
        torch.cuda.synchronize()
        want = float((st.grad.double() ** 2).sum())
        opt.step(max_grad_norm=1.0, zero_grad=True)
        got = float(opt.last_grad_norm_sq)
        assert got == pytest.approx(want, rel=1e-5), (i, got, want)
