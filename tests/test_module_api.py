"""CPU tests of the host-side mirror of the reference's module API: parameter names / shapes (checkpoint
compatibility), weight topology, flat-store layout, constructor errors.  No kernels run."""
import pytest
import torch

from oracle import reference_model as R


def _ours(cls="Seq2Seq", **kw):
    import imagetranslate_amd.seq2seq as S
    import imagetranslate_amd.image_model as I
    c = {"Seq2Seq": S.Seq2Seq, "ImageMassSeq2Seq": I.ImageMassSeq2Seq, "ImageCaptioning": I.ImageCaptioning}[cls]
    return c(R.SyntheticTextProcessor(1000), **kw)


@pytest.mark.parametrize("kw", [dict(lang_dec=False, enc_layer=2, dec_layer=2), dict(lang_dec=False, enc_layer=2, dec_layer=1),
                                dict(lang_dec=True, enc_layer=2, dec_layer=1), dict(lang_dec=False, tie_embed=True, enc_layer=1, dec_layer=1),
                                dict(lang_dec=True, tie_embed=True, enc_layer=1, dec_layer=1)])
def test_state_dict_keys_and_shapes_match_oracle(kw):
    kw = dict(kw, embed_dim=64, intermediate_dim=128, num_attention_heads=4)
    ref = R.Seq2Seq(R.SyntheticTextProcessor(1000), **kw)
    ours = _ours(**kw)
    a, b = ref.state_dict(), ours.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert a[k].shape == b[k].shape, k
    # same sharing pattern (which keys alias the same storage)
    def groups(sd):
        by = {}
        for k, v in sd.items():
            by.setdefault(v.data_ptr(), []).append(k)
        return sorted(sorted(v) for v in by.values() if len(v) > 1)
    assert groups(a) == groups(b)
    ours.load_state_dict(a)  # strict


def test_weight_topology_and_attributes():
    m = _ours(lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=64, intermediate_dim=128, num_attention_heads=4)
    assert m.decoder.decoder.layer[0].attention is m.encoder.encoder.layer[0].attention
    assert m.encoder.embeddings.word_embeddings.weight is m.decoder.embeddings.word_embeddings.weight
    assert m.encoder.embeddings.position_embeddings.num_embeddings == 512          # seq_gen.py:114
    assert m.config.vocab_size == 1000 and m.config.hidden_size == 64               # seq_gen.py:128
    assert hasattr(m.decoder.decoder.layer[0], "crossattention")
    assert not hasattr(m.encoder.encoder.layer[0], "crossattention")
    assert isinstance(m.output_layer, torch.nn.ModuleList) and len(m.output_layer) == 2
    assert m.lang_dec is False and m.tie_embed is False and m.use_proposals is False


def test_heads_must_divide_hidden():
    with pytest.raises(ValueError):
        _ours(embed_dim=512, intermediate_dim=1024, enc_layer=1, dec_layer=1)  # reference default 12 heads, 512 % 12 != 0


def test_flat_store_layout_on_cpu():
    from imagetranslate_amd.param_store import store_of
    m = _ours(lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=64, intermediate_dim=128, num_attention_heads=4)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    st = store_of(m.encoder).ensure()
    assert st is store_of(m.decoder)
    assert st.valid() and len(st.entries) == len(list(m.parameters()))
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k]), k
    a = m.encoder.encoder.layer[1].attention.self
    off = st.offset(a.query.weight)
    assert st.offset(a.key.weight) == off + 64 * 64 and st.offset(a.value.weight) == off + 2 * 64 * 64
    assert st.offset(a.key.bias) == st.offset(a.query.bias) + 64
    assert all(o % 64 == 0 for _, o, _ in st.entries)
    # gradient views alias the flat gradient buffer; zero_grad keeps them
    p = m.output_layer[0].layer.weight
    st.grad.fill_(1.0)
    assert float(p.grad.sum()) == p.numel()
    m.zero_grad()
    assert float(st.grad.abs().sum()) == 0 and p.grad is not None
    # in-place edits of a parameter land in the flat master
    with torch.no_grad():
        p.fill_(3.0)
    assert float(st.flat[st.offset(p)]) == 3.0


def test_image_models_construct_without_network():
    m = _ours("ImageMassSeq2Seq", enc_layer=1, dec_layer=1, embed_dim=64, intermediate_dim=128, num_attention_heads=4,
              resnet_depth=3)
    assert m.image_model.fc.in_features == 2048 and m.image_model.fc.bias is None
    assert m.image_model.location_embedding.num_embeddings == 49
    c = _ours("ImageCaptioning", enc_layer=1, dec_layer=1, embed_dim=64, intermediate_dim=128, num_attention_heads=4)
    assert hasattr(c, "obj_decoder") and hasattr(c, "multistream_attention_gate")


def test_optimizer_lr_schedule_matches_oracle():
    from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup
    p = torch.nn.Parameter(torch.zeros(3))
    opt = AdamInverseSqrtWithWarmup([p], lr=1e-4, betas=(0.9, 0.98), warmup_updates=5)
    for t in [0, 1, 4, 5, 6, 100, 10 ** 6]:
        assert opt.get_lr_for_step(t) == pytest.approx(R.inverse_sqrt_lr(t, 1e-4, 5))
    assert opt.param_groups[0]["lr"] == 1e-7
