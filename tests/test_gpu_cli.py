"""End-to-end on the GPU through the drop-in entry points: train a tokenizer on a tiny
synthetic parallel corpus, write the marshal example files, train with the trainer CLI, check the loss falls and the
best checkpoint reloads, then translate with beam search."""
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _corpus(n, seed):
    """'Language' A: random words; language B: the same sentence reversed with every word mapped w -> w + 'o'."""
    rnd = random.Random(seed)
    words = ["".join(rnd.choice("bcdfghklmnprst") + rnd.choice("aeiu") for _ in range(rnd.randint(1, 2))) for _ in range(60)]
    src, dst = [], []
    for _ in range(n):
        k = rnd.randint(3, 9)
        ws = [rnd.choice(words) for _ in range(k)]
        src.append(" ".join(ws))
        dst.append(" ".join(w + "o" for w in reversed(ws)))
    return src, dst


def test_tokenizer_batches_training_and_translation(cuda, tmp_path, capsys):
    from imagetranslate_amd import create_mt_batches, train_image_mt, train_tokenizer, translate
    from imagetranslate_amd.image_model import ImageMassSeq2Seq
    from imagetranslate_amd.seq_gen import BeamDecoder
    from imagetranslate_amd.textprocessor import TextProcessor
    src, dst = _corpus(1200, 5)
    d = str(tmp_path)
    with open(os.path.join(d, "all.txt"), "w") as fw:
        fw.write("\n".join(["<xa> " + s + " </s>" for s in src] + ["<xb> " + t + " </s>" for t in dst]) + "\n")
    for name, lines in (("train.xa", src[:1100]), ("train.xb", dst[:1100]), ("dev.xa", src[1100:]), ("dev.xb", dst[1100:])):
        with open(os.path.join(d, name), "w") as fw:
            fw.write("\n".join(lines) + "\n")
    tok = os.path.join(d, "tok")
    train_tokenizer.main(["--data", os.path.join(d, "all.txt"), "--vocab_size", "400", "--model", tok])
    tp = TextProcessor(tok)
    assert tp.languages == {"<xa>": 0, "<xb>": 1} and tp.pad_token_id() == 0 and tp.sep_token_id() == 4
    for split in ("train", "dev"):
        create_mt_batches.main(["--src", os.path.join(d, split + ".xa"), "--dst", os.path.join(d, split + ".xb"), "--src-lang", "xa",
                                "--dst-lang", "xb", "--tok", tok, "--output", os.path.join(d, split + ".batch")])
    model_dir = os.path.join(d, "model")
    opts, _ = train_image_mt.get_option_parser().parse_args(
        ["--train_mt", os.path.join(d, "train.batch"), "--dev_mt", os.path.join(d, "dev.batch"), "--tok", tok, "--model", model_dir,
         "--embed", "128", "--intermediate", "512", "--enc", "2", "--dec", "2", "--heads", "4", "--batch", "1500", "--capacity", "50",
         "--lr", "0.002", "--warmup", "40", "--step", "260", "--epoch", "40", "--eval-steps", "130", "--log-steps", "65", "--fp32"])
    trainer = train_image_mt.train(opts)
    log = capsys.readouterr().out
    losses = [float(ln.split("loss ")[1].split()[0]) for ln in log.splitlines() if " step " in ln and "loss " in ln]
    assert len(losses) >= 3 and losses[-1] < 0.6 * losses[0], losses
    assert os.path.exists(os.path.join(model_dir, "mt_model.state_dict")) and trainer.best_loss < losses[0]
    # reload (reference checkpoint layout + the heads extension) and translate the dev set
    model = ImageMassSeq2Seq.load(ImageMassSeq2Seq, model_dir, tok_dir=tok).cuda().eval()
    assert model.config.num_attention_heads == 4
    gen = BeamDecoder(model, beam_width=3, max_len_a=1.5, max_len_b=4)
    hyp = translate.translate_lines(model, gen, tp, src[1100:1140], tp.token_id("<xa>"), tp.token_id("<xb>"), max_tokens=600)
    assert len(hyp) == 40 and all(isinstance(h, str) for h in hyp)
    ref_words = [set(t.split()) for t in dst[1100:1140]]
    overlap = sum(len(set(h.split()) & r) / max(len(r), 1) for h, r in zip(hyp, ref_words)) / 40
    assert overlap > 0.3, "after 260 steps the toy model should reproduce a good share of the mapped words (got %.2f)" % overlap
    # file interface of the CLI
    translate.main(["--input", os.path.join(d, "dev.xa"), "--output", os.path.join(d, "dev.out"), "--src", "xa", "--target", "xb",
                    "--tok", tok, "--model", model_dir, "--beam", "2", "--fp32"])
    with open(os.path.join(d, "dev.out")) as fp:
        assert len(fp.read().strip().split("\n")) == 100


def test_trainer_mass_steps_on_device_batches(cuda):
    """MASS objective through the trainer: MassDataset batches -> imt_mass_mask on the device -> decoder with the
    original positions -> fused loss; the loss falls on a tiny memorisable corpus."""
    from imagetranslate_amd.dataset import MassDataset
    from imagetranslate_amd.image_model import ImageMassSeq2Seq
    from imagetranslate_amd.textprocessor import SyntheticTextProcessor
    from imagetranslate_amd.train_image_mt import ImageMTTrainer
    from imagetranslate_amd.utils import build_optimizer
    rnd = random.Random(2)
    torch.manual_seed(2)
    random.seed(2)
    tp = SyntheticTextProcessor(300)
    sents = [[5] + [rnd.randint(7, 60) for _ in range(rnd.randint(8, 20))] + [4] for _ in range(48)]
    sents.sort(key=len)
    data = MassDataset(None, max_batch_capacity=50, max_batch=700, pad_idx=0, max_seq_len=64, example_list=[[(s, 0) for s in sents]])
    assert len(data) >= 2 and all(b["src_texts"].size(0) >= 1 for b in data.batches)
    model = ImageMassSeq2Seq(tp, lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512,
                             num_attention_heads=4).cuda().train()
    trainer = ImageMTTrainer(model, mask_prob=0.3, optimizer=build_optimizer(model, 2e-3, 20))
    losses = []
    for epoch in range(25):
        for b in data.batches:
            before = b["src_texts"].clone()
            loss, n = trainer.mass_step(b)
            assert torch.equal(b["src_texts"], before), "the dataset's tensors are not modified"
            assert n == int((b["pad_idx"] // 2).sum())
            losses.append(float(loss))
    k = len(data)
    assert sum(losses[-k:]) / k < 0.75 * sum(losses[:k]) / k, (losses[:k], losses[-k:])


def test_captioning_train_and_caption_cli(cuda, tmp_path, capsys):
    """train_captioning.py / caption.py counterparts end to end on the HIP path: region-feature files in place of pixels,
    image batches + MT batches as a second task weighted by --mtlw, --acc 2, then beam-search captions from the saved
    checkpoint.  Image i's caption is a fixed function of its features' class, so the loss must fall."""
    import marshal
    from imagetranslate_amd import caption, create_mt_batches, train_captioning, train_tokenizer
    from imagetranslate_amd.textprocessor import TextProcessor
    d = str(tmp_path)
    rnd = random.Random(3)
    src, dst = _corpus(300, 9)
    with open(os.path.join(d, "all.txt"), "w") as fw:
        fw.write("\n".join(["<xa> " + s + " </s>" for s in src] + ["<xb> " + t + " </s>" for t in dst]) + "\n")
    tok = os.path.join(d, "tok")
    train_tokenizer.main(["--data", os.path.join(d, "all.txt"), "--vocab_size", "300", "--model", tok])
    tp = TextProcessor(tok)
    # 8 image classes; the caption of an image is the class's sentence
    torch.manual_seed(0)
    protos = torch.randn(8, 49, 64)
    class_caps = [src[i] for i in range(8)]
    n_img = 96
    labels = [rnd.randrange(8) for _ in range(n_img)]
    feats = torch.stack([protos[c] + 0.05 * torch.randn(49, 64) for c in labels])
    paths = ["img%03d.jpg" % i for i in range(n_img)]
    img_dir = os.path.join(d, "images")
    os.makedirs(img_dir)
    torch.save({"paths": paths, "feats": feats}, os.path.join(img_dir, "features.pt"))
    unique = {i: paths[i] for i in range(n_img)}
    caps = [(i, tp.tokenize_one_sentence_with_langid(class_caps[labels[i]], tp.token_id("<xa>"))) for i in range(n_img)]
    caps.sort(key=lambda c: len(c[1]))
    for name, part in (("train.cap", caps[:80]), ("dev.cap", caps[80:])):
        with open(os.path.join(d, name), "wb") as fw:
            marshal.dump((unique, part), fw)
    with open(os.path.join(d, "mt.xa"), "w") as fw:
        fw.write("\n".join(src[:200]) + "\n")
    with open(os.path.join(d, "mt.xb"), "w") as fw:
        fw.write("\n".join(dst[:200]) + "\n")
    create_mt_batches.main(["--src", os.path.join(d, "mt.xb"), "--dst", os.path.join(d, "mt.xa"), "--src-lang", "xb", "--dst-lang", "xa",
                            "--tok", tok, "--output", os.path.join(d, "mt.batch")])
    model_dir = os.path.join(d, "cap_model")
    train_captioning.main(["--train", os.path.join(d, "train.cap"), "--dev", os.path.join(d, "dev.cap"), "--image", img_dir,
                           "--train_mt", os.path.join(d, "mt.batch"), "--tok", tok, "--model", model_dir, "--embed", "128",
                           "--intermediate", "256", "--enc", "1", "--dec", "2", "--heads", "4", "--feat-dim", "64", "--no-obj",
                           "--max-image", "16", "--batch", "1200", "--lr", "0.003", "--warmup", "20", "--step", "240", "--epoch", "60",
                           "--acc", "2", "--mtlw", "0.1", "--log-steps", "40", "--eval-steps", "120", "--fp32"])
    log = capsys.readouterr().out
    losses = [float(ln.split("Loss: ")[1].split()[0]) for ln in log.splitlines() if "Epoch Step" in ln]
    assert len(losses) >= 4 and losses[-1] < 0.7 * losses[0], losses
    assert os.path.exists(os.path.join(model_dir, "mt_model.state_dict"))
    out_file = os.path.join(d, "captions.txt")
    caption.main(["--input", img_dir, "--target", "xa", "--output", out_file, "--tok", tok, "--model", model_dir, "--beam", "2",
                  "--batch", "32", "--max-len", "24", "--fp32"])
    lines = open(out_file).read().strip().split("\n")
    assert len(lines) == n_img and all("\t" in ln for ln in lines)
    by_path = dict(ln.split("\t", 1) for ln in lines)
    hit = sum(len(set(by_path[paths[i]].split()) & set(class_caps[labels[i]].split())) / max(1, len(set(class_caps[labels[i]].split())))
              for i in range(n_img)) / n_img
    assert hit > 0.3, "captions should reproduce a good share of the class sentences' words (got %.2f)" % hit


def test_config0_sample_corpus_against_oracle(cuda, tmp_path):
    """BASELINE configs[0]: "MT on sample/ en<->fa toy pairs, 2-layer d=128 seq_len=32 batch=8" -- on the reference's own toy
    corpus (tests/golden/sample_enfa/: the first 400 line pairs of src/sample/{en,fa}.txt, README.md:167's smoke data): BPE
    tokenizer (vocabulary 1000, README.md:147), create_mt_batches, capacity batching, then five train steps (forward, smoothed
    NLL, backward, clip, Adam with the inverse-sqrt schedule) of the HIP path in fp32 against the CPU oracle on the same
    weights and batches: per-step losses within 1e-4, an updated weight within 1e-4, and the loss falls."""
    from imagetranslate_amd import create_mt_batches, train_tokenizer
    from imagetranslate_amd.dataset import MTDataset
    from imagetranslate_amd.parallel import train_step
    from imagetranslate_amd.seq2seq import Seq2Seq
    from imagetranslate_amd.textprocessor import TextProcessor
    from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup
    from oracle import reference_model as R
    gold = os.path.join(os.path.dirname(__file__), "golden", "sample_enfa")
    d = str(tmp_path)
    en = [ln.strip() for ln in open(os.path.join(gold, "en.txt"), encoding="utf-8")]
    fa = [ln.strip() for ln in open(os.path.join(gold, "fa.txt"), encoding="utf-8")]
    assert len(en) == len(fa) == 400
    with open(os.path.join(d, "all.txt"), "w", encoding="utf-8") as fw:
        fw.write("\n".join(["<en> " + s + " </s>" for s in en if s] + ["<fa> " + t + " </s>" for t in fa if t]) + "\n")
    tok = os.path.join(d, "tok")
    train_tokenizer.main(["--data", os.path.join(d, "all.txt"), "--vocab_size", "1000", "--model", tok])
    tp = TextProcessor(tok)
    assert tp.languages == {"<en>": 0, "<fa>": 1} and 600 <= tp.vocab_size() <= 1000
    create_mt_batches.main(["--src", os.path.join(gold, "en.txt"), "--dst", os.path.join(gold, "fa.txt"), "--src-lang", "en", "--dst-lang", "fa",
                            "--tok", tok, "--output", os.path.join(d, "train.batch"), "--max_seq_len", "32", "--min_seq_len", "3"])
    data = MTDataset(max_batch_capacity=600, max_batch=8 * 64, pad_idx=tp.pad_token_id(), max_seq_len=32, batch_pickle_dir=os.path.join(d, "train.batch"))
    batches = [b for b in data.batches if 4 <= b["src_texts"].size(0) <= 8][:5]
    assert len(batches) == 5 and all(b["src_texts"].size(1) <= 32 and b["dst_texts"].size(1) <= 32 for b in batches)
    kw = dict(lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512, num_attention_heads=4)
    torch.manual_seed(7)
    ref = R.Seq2Seq(tp, **kw).eval()          # dropout off on both sides (parity run)
    ours = Seq2Seq(tp, **kw)
    ours.load_state_dict(ref.state_dict())
    ours = ours.cuda().eval()
    ours.set_compute_dtype(torch.float32)
    opt_r = R.AdamInverseSqrtWithWarmup(ref.parameters(), lr=2e-3, betas=(0.9, 0.98), warmup_updates=3)
    opt_o = AdamInverseSqrtWithWarmup(ours.parameters(), lr=2e-3, betas=(0.9, 0.98), warmup_updates=3)
    crit = R.SmoothedNLLLoss(ignore_index=tp.pad_token_id())
    lr_, lo_ = [], []
    for rep in range(2):
        for b in batches:
            lr_.append(R.train_step(ref, opt_r, crit, b)[0])
            loss, ntok = train_step(ours, opt_o, b)
            lo_.append(float(loss))
            assert ntok == int(b["dst_pad_mask"][:, 1:].sum())
    for a, b_ in zip(lo_[:5], lr_[:5]):
        assert a == pytest.approx(b_, rel=1e-4), (lo_, lr_)
    assert lo_ == pytest.approx(lr_, rel=2e-3), "ten optimizer steps stay on the oracle's trajectory"
    assert sum(lo_[5:]) < sum(lo_[:5]), "second pass over the same batches must have a lower loss"
    k = "decoder.decoder.layer.1.crossattention.self.query.weight"
    wr, wo = dict(ref.named_parameters())[k].detach(), dict(ours.named_parameters())[k].detach().cpu()
    assert float((wr - wo).abs().max() / wr.abs().max()) < 2e-3


def test_back_translation_step_against_oracle(cuda):
    """The back-translation phase of the trainer (src/train_image_mt.py:108-198, monolingual batches): the model translates the
    batch into the other language (greedy, no gradient), then takes a train step on (translation -> original).  Against the
    oracle on the same weights: the generated token ids bit-exact, the loss within 1e-4, the target count equal; and the
    command-line plumbing (--langs / --fstep / --bt-beam) resolves the language pair."""
    from imagetranslate_amd.image_model import ImageMassSeq2Seq
    from imagetranslate_amd.option_parser import get_img_options_parser
    from imagetranslate_amd.textprocessor import SyntheticTextProcessor
    from imagetranslate_amd.train_image_mt import ImageMTTrainer, reject_off_path
    from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup
    from oracle import reference_model as R
    from oracle.seq_gen import BeamDecoder as OracleBeam
    from torch.nn.utils.rnn import pad_sequence
    from tests.util import beam_state_dict
    fx = torch.load(os.path.join(os.path.dirname(__file__), "golden", "toy_seq2seq.pt"), weights_only=True)
    kw = dict(lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512, num_attention_heads=4)
    ref = R.Seq2Seq(R.SyntheticTextProcessor(1000), **kw)
    ref.load_state_dict(beam_state_dict(fx["state_dict"]))
    ref.eval()
    tp = SyntheticTextProcessor(1000)
    ours = ImageMassSeq2Seq(tp, **kw)
    missing = ours.load_state_dict(ref.state_dict(), strict=False)
    assert not missing.unexpected_keys
    ours = ours.cuda().eval()
    ours.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(12)
    B, S = 6, 14
    src = torch.randint(7, 1000, (B, S), generator=g)
    lens = torch.tensor([14, 9, 12, 7, 14, 10])
    src[:, 0] = torch.tensor([5, 6, 5, 5, 6, 6])           # language tags <en> = 5, <fa> = 6 first
    for b in range(B):
        src[b, lens[b] - 1] = 4
        src[b, lens[b]:] = 0
    langs = (src[:, 0] - 5).clone()                          # language ids 0 / 1
    batch = {"src_texts": src, "langs": langs, "pad_idx": torch.where(lens < S, lens, torch.full_like(lens, S - 1))}
    dirs = ImageMTTrainer.get_lang_dirs("en,fa", tp)
    assert dirs == {5: 6, 6: 5}
    # expectation from the oracle, per one-language batch (as MassDataset builds them, src/dataset.py:212-269): greedy
    # translations, then the smoothed NLL of (translation -> original); the reference's forward takes ONE target language per batch
    from imagetranslate_amd.seq_gen import BeamDecoder
    exp = []
    for lang in (0, 1):
        rows = (langs == lang).nonzero().view(-1)
        tags = torch.LongTensor([dirs[int(t)] for t in src[rows, 0]])
        dst_langs = tags - 5
        gen_kw = dict(src_inputs=src[rows], src_sizes=batch["pad_idx"][rows], first_tokens=tags, src_langs=langs[rows], tgt_langs=dst_langs,
                      pad_idx=0, src_mask=src[rows] != 0, unpad_output=False)
        outs = OracleBeam(ref, beam_width=1, max_len_a=1.3, max_len_b=5)(**gen_kw)
        got = BeamDecoder(ours, beam_width=1, max_len_a=1.3, max_len_b=5)(**gen_kw)
        assert [o.tolist() for o in got] == [o.tolist() for o in outs], "back-translations must be the oracle's token ids"
        trans = pad_sequence(outs, batch_first=True, padding_value=0)
        lp = ref(trans, src[rows], trans != 0, src[rows] != 0, dst_langs, langs[rows], log_softmax=True)
        tg = src[rows][:, 1:][(src[rows] != 0)[:, 1:]]
        exp.append((float(R.SmoothedNLLLoss(ignore_index=0)(lp, tg).sum().detach()), int(tg.numel())))
    opt = AdamInverseSqrtWithWarmup(ours.parameters(), lr=1e-4, betas=(0.9, 0.98), warmup_updates=10)
    trainer = ImageMTTrainer(ours, optimizer=opt, clip=1.0, bt_beam_width=1)
    w0 = dict(ours.named_parameters())["decoder.decoder.layer.0.output.dense.weight"].detach().clone()
    for lang in (0, 1):   # one-language batches, as MassDataset builds them (src/dataset.py:212-269 groups by language)
        rows = (langs == lang).nonzero().view(-1)
        sub = {"src_texts": src[rows], "langs": langs[rows], "pad_idx": batch["pad_idx"][rows]}
        loss, n = trainer.bt_step(sub, dirs)
        assert n == exp[lang][1]
        assert float(loss) == pytest.approx(exp[lang][0] / exp[lang][1], rel=2e-4 if lang == 0 else 5e-3)  # (second step: weights moved once)
    w1 = dict(ours.named_parameters())["decoder.decoder.layer.0.output.dense.weight"].detach()
    assert float((w1 - w0).abs().max()) > 0
    o, _ = get_img_options_parser().parse_args(["--langs", "en,fa", "--fstep", "20", "--bt-beam", "1"])
    reject_off_path(o)   # accepted since round 3
    assert (o.bt_langs, o.finetune_step, o.bt_beam_width) == ("en,fa", 20, 1)
