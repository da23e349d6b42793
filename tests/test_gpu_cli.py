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
        ["--train", os.path.join(d, "train.batch"), "--dev", os.path.join(d, "dev.batch"), "--tok", tok, "--model", model_dir,
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
            losses.append(loss)
    k = len(data)
    assert sum(losses[-k:]) / k < 0.75 * sum(losses[:k]) / k, (losses[:k], losses[-k:])
