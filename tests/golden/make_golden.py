"""Generates the committed golden fixtures.  Run in the BUILD container only (needs /root/reference):

    python tests/golden/make_golden.py

* loss_kat.json : known-answer vectors from the REFERENCE's own src/loss.py (SmoothedNLLLoss), imported from
                  /root/reference/src -- inputs + expected per-row loss + d(mean loss)/dlogits through
                  F.log_softmax.  This is data (inputs/outputs), no reference source is copied.
* toy_seq2seq.pt: whole-model vectors from THIS repo's oracle (oracle/reference_model.py) on the toy
                  configuration of SURVEY.md section 8(c) (d=128, h=4, ff=512, 2+2 layers, V=1000, B=8, S=T=32,
                  dropout off, seed fixed): inputs, state_dict, encoder states, log-probs, loss, selected grads,
                  three optimizer steps.  The reference's own tests pin nothing numeric for the model
                  (parity unpinned) -- this fixture guards the oracle against drift.
* beam_kat.json : known-answer vectors from the REFERENCE's own src/seq_gen.py get_outputs_until_eos (imported
                  from /root/reference/src): token matrices + eos / size_limit / remove_first_token -> cut rows.
* config_kat.json: known answers from the REFERENCE's own src/lm_config.py (get_config dicts for three model sizes) and
                  src/textprocessor.py (the special-token id layout of a tokenizer it trains: pad=0, <s>=1, <unk>=2,
                  <mask>=3, </s>=4, language tags next; framing of a "<lang> ... </s>" line), both imported from
                  /root/reference/src.
* toy_beam.pt   : beam-search token ids from THIS repo's oracle (oracle/seq_gen.py) on the toy model derived from
                  toy_seq2seq.pt by tests/util.py:beam_state_dict (matrices x2, EOS bias) so that hypotheses finish
                  at different steps; text (beam 1 / 4, with and without unpadding) and image-only captioning.
                  The reference has no test or fixture for beam search (parity unpinned).
* beam_step1_kat.pt: the REFERENCE's own src/seq_gen.py BeamDecoder.forward run un-modified (imported from /root/reference/src)
                  on top of THIS repo's oracle model with max_len=2 -- only iteration i = 1 executes, which is the part of
                  the loop that runs on torch >= 1.5 (the true division of :216 sits under ``if i > 1``): encode, first-token
                  handling, decoder call, log_softmax, EOS zeroing, length penalty, top-k over V, the cat of the chosen word
                  (src/seq_gen.py:94-131,133-203,225-233).  Recorded: the arguments / results of its ``torch.topk`` call and
                  the token rows it returns, for beam 1 / 4 / 5 on the text path and beam 3 on the ``images=`` path.
* marshal_kat/  : the REFERENCE's own src/create_mt_batches.py ``write()`` (imported) on a 50-line parallel corpus and on
                  the monolingual source side, with a tokenizer its own src/textprocessor.py trains and re-loads:
                  corpus, tokenizer files (vocab.json / merges.txt / langs) and the two marshal files it wrote.
* sample_enfa/   : the first 400 line pairs of the REFERENCE's own toy corpus src/sample/en.txt / fa.txt (BASELINE configs[0]: "MT on
                  sample/ en<->fa toy pairs"), data only -- what README.md:167 tells a user to smoke-test on.
* options_kat.json: every option of the REFERENCE's own src/option_parser.py parsers (flag strings, dest, type, action,
                  default), read from the parser objects it builds.
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def make_loss_kat():
    sys.path.insert(0, "/root/reference/src")
    import loss as ref_loss  # the reference's own file
    g = torch.Generator().manual_seed(20240)
    cases = []
    specs = [
        (torch.tensor([[1., 2., 3., 4.], [.5, -.5, 0., 2.], [0., 0., 0., 0.]]), torch.tensor([3, 1, 0]), 0.1, 0),
        (torch.randn(6, 16, generator=g) * 2, torch.randint(0, 16, (6,), generator=g), 0.1, 0),
        (torch.randn(5, 24, generator=g), torch.randint(1, 24, (5,), generator=g), 0.2, -100),
    ]
    for logits, tgt, eps, ign in specs:
        z = logits.clone().requires_grad_(True)
        lp = F.log_softmax(z, dim=-1)
        crit = ref_loss.SmoothedNLLLoss(ignore_index=ign, epsilon=eps)
        l = crit(lp, tgt)
        l.mean().backward()
        cases.append({"logits": logits.tolist(), "target": tgt.tolist(), "epsilon": eps, "ignore_index": ign,
                      "loss": l.detach().view(-1).tolist(), "dlogits_mean": z.grad.tolist()})
    json.dump({"source": "rasoolims/ImageTranslate src/loss.py SmoothedNLLLoss via F.log_softmax", "cases": cases},
              open(os.path.join(HERE, "loss_kat.json"), "w"), indent=1)


def make_toy():
    from oracle import reference_model as R
    torch.manual_seed(1234)
    tp = R.SyntheticTextProcessor(1000)
    m = R.Seq2Seq(tp, lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512,
                  num_attention_heads=4)
    m.eval()  # dropout off
    B, S, T = 8, 32, 32
    g = torch.Generator().manual_seed(99)
    src = torch.randint(6, 1000, (B, S), generator=g)
    tgt = torch.randint(6, 1000, (B, T), generator=g)
    ls = torch.randint(16, S + 1, (B,), generator=g); lt = torch.randint(16, T + 1, (B,), generator=g)
    src[torch.arange(S)[None] >= ls[:, None]] = 0
    tgt[torch.arange(T)[None] >= lt[:, None]] = 0
    src[:, 0] = 5; tgt[:, 0] = 6
    batch = {"src_texts": src, "dst_texts": tgt, "src_pad_mask": src != 0, "dst_pad_mask": tgt != 0,
             "src_langs": torch.zeros(B, dtype=torch.long), "dst_langs": torch.ones(B, dtype=torch.long)}
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    enc = m.encode(src, src != 0, batch["src_langs"].unsqueeze(-1).expand(-1, S))[0]
    lp = m(src, tgt, src != 0, tgt != 0, batch["src_langs"], batch["dst_langs"], log_softmax=True)
    targets = tgt[:, 1:].contiguous().view(-1)[(tgt != 0)[:, 1:].contiguous().view(-1)]
    crit = R.SmoothedNLLLoss(ignore_index=0)
    loss = crit(lp, targets).mean()
    loss.backward()
    named = dict(m.named_parameters())
    grad_keys = ["encoder.embeddings.word_embeddings.weight",
                 "encoder.encoder.layer.0.attention.self.query.weight",
                 "decoder.decoder.layer.1.crossattention.self.key.weight",
                 "output_layer.1.layer.weight", "output_layer.1.layer.bias"]
    grads = {k: named[k].grad.clone() for k in grad_keys}
    m.zero_grad()
    opt = R.AdamInverseSqrtWithWarmup(m.parameters(), lr=1e-3, betas=(0.9, 0.98), warmup_updates=2)
    losses = [R.train_step(m, opt, crit, batch)[0] for _ in range(3)]
    torch.save({"batch": batch, "state_dict": sd0, "encoder_states": enc.detach(), "log_probs": lp.detach(),
                "argmax": lp.argmax(-1), "loss": float(loss), "grads": grads, "step_losses": losses,
                "updated_query_weight": named["encoder.encoder.layer.0.attention.self.query.weight"].detach().clone(),
                "config": dict(vocab=1000, enc=2, dec=2, d=128, ff=512, heads=4, lr=1e-3, warmup=2)},
               os.path.join(HERE, "toy_seq2seq.pt"))


def make_beam_kat():
    sys.path.insert(0, "/root/reference/src")
    import seq_gen as ref_gen  # the reference's own file (needs only torch)
    g = torch.Generator().manual_seed(7)
    cases = []
    for rows, cols, eos, with_limit, rm_first in [(5, 9, 4, False, False), (6, 12, 4, True, False), (4, 7, 2, True, True),
                                                  (1, 5, 4, False, True)]:
        m = torch.randint(0, 8, (rows, cols), generator=g)
        limit = torch.randint(2, cols + 1, (rows,), generator=g) if with_limit else None
        out = ref_gen.get_outputs_until_eos(eos, m, size_limit=limit, remove_first_token=rm_first)
        cases.append({"outputs": m.tolist(), "eos": eos, "size_limit": None if limit is None else limit.tolist(),
                      "remove_first_token": rm_first, "expected": [o.tolist() for o in out]})
    v = torch.tensor([5, 9, 4, 7, 4])
    cases.append({"outputs": v.tolist(), "eos": 4, "size_limit": None, "remove_first_token": False,
                  "expected": [o.tolist() for o in ref_gen.get_outputs_until_eos(4, v)]})
    json.dump({"source": "rasoolims/ImageTranslate src/seq_gen.py get_outputs_until_eos", "cases": cases},
              open(os.path.join(HERE, "beam_kat.json"), "w"), indent=1)


def make_beam():
    from oracle import reference_model as R
    from oracle.seq_gen import BeamDecoder
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import beam_inputs, beam_state_dict, caption_beam_inputs
    fx = torch.load(os.path.join(HERE, "toy_seq2seq.pt"), weights_only=True)
    tp = R.SyntheticTextProcessor(1000)
    m = R.Seq2Seq(tp, lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512, num_attention_heads=4)
    m.load_state_dict(beam_state_dict(fx["state_dict"]))
    m.eval()
    inp = beam_inputs()
    res = {}
    for name, kw in [("beam4", dict(beam_width=4)), ("beam1", dict(beam_width=1)), ("beam3_padded", dict(beam_width=3, unpad_output=False)),
                     ("beam4_maxlen10", dict(beam_width=4, max_len=10))]:
        trace = []
        out = BeamDecoder(m, beam_width=5)(pad_idx=0, trace=trace, **inp, **kw)
        res[name] = {"tokens": [o.clone() for o in out], "steps": len(trace),
                     "trace_outs": [t["outs"].clone() for t in trace]}
    m.load_state_dict(beam_state_dict(fx["state_dict"], 2.5, 2.5))
    res["beam4_soft"] = {"tokens": [o.clone() for o in BeamDecoder(m, beam_width=4)(pad_idx=0, **inp)]}
    torch.manual_seed(4321)
    cap = R.ImageCaptioning(tp, lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512,
                            num_attention_heads=4, image_feat_dim=64)
    extra = {k: v.clone() for k, v in cap.state_dict().items() if k not in fx["state_dict"]}
    cap.load_state_dict({**beam_state_dict(fx["state_dict"]), **extra})
    cap.eval()
    out = BeamDecoder(cap, beam_width=3)(pad_idx=0, max_len=14, **caption_beam_inputs())
    res["caption_beam3"] = {"tokens": [o.clone() for o in out], "extra_state": extra}
    torch.save(res, os.path.join(HERE, "toy_beam.pt"))
    for k, v in res.items():
        print(k, [len(t) for t in v["tokens"]])


def make_config_kat():
    """Known answers from the REFERENCE's own src/lm_config.py (get_config) and src/textprocessor.py (special-token id layout
    of a tokenizer it trains itself) -- both import in the build container.  Data only: the config dicts it returns and the
    ids it assigns."""
    sys.path.insert(0, "/root/reference/src")
    import lm_config as ref_cfg
    import tempfile
    out = {"source": "rasoolims/ImageTranslate src/lm_config.py get_config, src/textprocessor.py TextProcessor", "configs": []}
    for args in [dict(vocab_size=30000, pad_token_id=0, bos_token_id=1, eos_token_id=4, enc_layer=6, embed_dim=768, intermediate_dim=3072),
                 dict(vocab_size=1000, pad_token_id=0, bos_token_id=1, eos_token_id=4, enc_layer=2, embed_dim=128, intermediate_dim=512),
                 dict(vocab_size=60000, pad_token_id=0, bos_token_id=1, eos_token_id=4, enc_layer=3, embed_dim=512, intermediate_dim=2048)]:
        out["configs"].append({"args": args, "config": ref_cfg.get_config(**args)})
    import textprocessor as ref_tp
    tp = ref_tp.TextProcessor()
    corpus = os.path.join(ROOT, "tests", "golden", "_tok_corpus.txt")
    words = ["alpha", "beta", "gamma", "delta", "omega", "sigma", "kappa", "theta"]
    with open(corpus, "w") as fw:
        for i in range(400):
            tag = "<en>" if i % 2 == 0 else "<fa>"
            fw.write(tag + " " + " ".join(words[(i * 3 + j) % 8] for j in range(3 + i % 5)) + " </s>\n")
    with tempfile.TemporaryDirectory() as d:
        try:
            tp.train_tokenizer(paths=[corpus], vocab_size=120, to_save_dir=d, languages={"<en>": 0, "<fa>": 1})
        except Exception as e:  # TextProcessor.save fails under tokenizers 0.22 ("Is a directory"); training itself has run
            out["save_error"] = type(e).__name__
    ids = {t: tp.tokenizer.token_to_id(t) for t in tp.special_tokens}
    line = "<en> alpha beta gamma </s>"
    out["textprocessor"] = {"special_tokens": list(tp.special_tokens), "special_ids": ids,
                            "pad": tp.pad_token_id(), "bos": tp.bos_token_id(), "unk": tp.unk_token_id(), "mask": tp.mask_token_id(),
                            "sep": tp.sep_token_id(), "languages": tp.languages, "corpus_seed_words": words,
                            "framed_line": line, "framed_first_last": [tp.tokenize_one_sentence(line)[0], tp.tokenize_one_sentence(line)[-1]],
                            "is_lang": {str(i): bool(tp.is_lang(i)) for i in range(8)}}
    os.remove(corpus)
    json.dump(out, open(os.path.join(HERE, "config_kat.json"), "w"), indent=1)


def _ref_beam_step1(ref_gen, model, beam, **inp):
    """Runs the reference's BeamDecoder.forward as it is, with max_len=2 (one iteration), recording its torch.topk call."""
    rec = {}
    real_topk = torch.topk

    def spy(x, *a, **kw):
        out = real_topk(x, *a, **kw)
        rec["scores_in"] = x.detach().clone()
        rec["top_scores"], rec["indices"] = out[0].detach().clone(), out[1].detach().clone()
        return out
    torch.topk = spy
    try:
        with torch.no_grad():
            toks = ref_gen.BeamDecoder(model, beam_width=beam)(pad_idx=0, max_len=2, **inp)
            padded = ref_gen.BeamDecoder(model, beam_width=beam)(pad_idx=0, max_len=2, unpad_output=False, **inp)
    finally:
        torch.topk = real_topk
    return {"beam": beam, "tokens": [t.clone() for t in toks], "tokens_padded": [t.clone() for t in padded],
            "top_scores": rec["top_scores"], "indices": rec["indices"],
            "row_logsumexp_check": torch.logsumexp(rec["scores_in"], 1) if beam == 1 else None}


def make_beam_step1_kat():
    sys.path.insert(0, "/root/reference/src")
    import seq_gen as ref_gen  # the reference's own file, un-modified
    from oracle import reference_model as R
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import beam_inputs, beam_state_dict, caption_beam_inputs
    fx = torch.load(os.path.join(HERE, "toy_seq2seq.pt"), weights_only=True)
    gold_beam = torch.load(os.path.join(HERE, "toy_beam.pt"), weights_only=True)
    tp = R.SyntheticTextProcessor(1000)
    kw = dict(lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=512, num_attention_heads=4)
    m = R.Seq2Seq(tp, **kw)
    m.load_state_dict(beam_state_dict(fx["state_dict"]))
    m.eval()
    res = {"source": "rasoolims/ImageTranslate src/seq_gen.py BeamDecoder.forward(max_len=2) on oracle/reference_model.py"}
    for beam in (1, 4, 5):
        res["text_beam%d" % beam] = _ref_beam_step1(ref_gen, m, beam, **beam_inputs())
    # a first token that IS the end-of-sentence id: rows zeroed by :194-195, ties in top-k (which order torch.topk gives them
    # in is recorded, not assumed)
    inp = beam_inputs()
    inp["first_tokens"] = torch.tensor([5, 4, 5, 4, 5, 5])
    res["text_beam4_eos_first"] = _ref_beam_step1(ref_gen, m, 4, **inp)
    cap = R.ImageCaptioning(tp, image_feat_dim=64, **kw)
    cap.load_state_dict({**beam_state_dict(fx["state_dict"]), **gold_beam["caption_beam3"]["extra_state"]})
    cap.eval()
    res["caption_beam3"] = _ref_beam_step1(ref_gen, cap, 3, **caption_beam_inputs())
    torch.save(res, os.path.join(HERE, "beam_step1_kat.pt"))
    for k, v in res.items():
        if isinstance(v, dict):
            print(k, [t.tolist() for t in v["tokens"]][:3])


def make_marshal_kat():
    """Reference create_mt_batches.write() -> marshal fixtures (data: corpus, tokenizer vocabulary, the bytes it wrote)."""
    import pickle
    import random
    import shutil
    sys.path.insert(0, "/root/reference/src")
    import create_mt_batches as ref_cmb
    import textprocessor as ref_tp
    out = os.path.join(HERE, "marshal_kat")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "tok"))
    rnd = random.Random(50)
    en = ["the", "cat", "dog", "sees", "a", "red", "house", "and", "runs", "to", "green", "tree", "under", "sun", "quickly"]
    fa = ["in", "gorbe", "sag", "mibinad", "yek", "ghermez", "khane", "va", "midavad", "be", "sabz", "derakht", "zire", "aftab", "tond"]
    with open(os.path.join(out, "src.txt"), "w") as fs, open(os.path.join(out, "dst.txt"), "w") as fd:
        for i in range(50):
            n, k = rnd.randint(1, 14), rnd.randint(1, 14)
            fs.write(" ".join(rnd.choice(en) for _ in range(n)) + "\n")
            fd.write(("" if i == 17 else " ".join(rnd.choice(fa) for _ in range(k))) + "\n")  # line 17: empty target -> skipped
    train_txt = os.path.join(out, "_train.txt")
    with open(train_txt, "w") as fw:
        for _ in range(20):
            fw.write(open(os.path.join(out, "src.txt")).read())
            fw.write(open(os.path.join(out, "dst.txt")).read())
    tp = ref_tp.TextProcessor()
    try:
        tp.train_tokenizer(paths=[train_txt], vocab_size=90, to_save_dir=os.path.join(out, "tok"), languages={"<en>": 0, "<fa>": 1})
    except Exception as e:  # TextProcessor.save: tokenizers 0.22 refuses a directory; the training itself has run
        print("reference TextProcessor.save failed as SURVEY 2#13 says:", type(e).__name__)
    os.remove(train_txt)
    for f in os.listdir(os.path.join(out, "tok")):
        os.remove(os.path.join(out, "tok", f))
    tp.tokenizer.save_model(os.path.join(out, "tok"))  # vocab.json + merges.txt, the layout its __init__ reads
    with open(os.path.join(out, "tok", "langs"), "wb") as fp:
        pickle.dump(tp.languages, fp)  # what its save() writes next (src/textprocessor.py:44-45)
    tp = ref_tp.TextProcessor(os.path.join(out, "tok"))  # re-loaded the reference's way
    src_lang, dst_lang = tp.token_id("<en>"), tp.token_id("<fa>")
    ref_cmb.write(text_processor=tp, output_file=os.path.join(out, "mt.marshal"), src_txt_file=os.path.join(out, "src.txt"),
                  src_lang=src_lang, dst_txt_file=os.path.join(out, "dst.txt"), dst_lang=dst_lang, min_len=3, max_len=30)
    ref_cmb.write(text_processor=tp, output_file=os.path.join(out, "mass.marshal"), src_txt_file=os.path.join(out, "src.txt"),
                  src_lang=src_lang, min_len=1, max_len=175)
    import marshal
    ex = marshal.load(open(os.path.join(out, "mt.marshal"), "rb"))
    mono = marshal.load(open(os.path.join(out, "mass.marshal.0"), "rb"))
    json.dump({"source": "rasoolims/ImageTranslate src/create_mt_batches.py write(), src/textprocessor.py",
               "src_lang_id": src_lang, "dst_lang_id": dst_lang, "n_parallel": len(ex), "n_mono": len(mono),
               "min_len_parallel": 3, "max_len_parallel": 30, "vocab_size": tp.tokenizer.get_vocab_size(),
               "sample_line": "the cat sees a red house", "sample_ids": tp.tokenize_one_sentence_with_langid("the cat sees a red house", src_lang)},
              open(os.path.join(out, "meta.json"), "w"), indent=1)
    print("marshal_kat:", len(ex), "parallel,", len(mono), "monolingual examples; vocab", tp.tokenizer.get_vocab_size())


def make_sample_excerpt(n=400):
    out = os.path.join(HERE, "sample_enfa")
    os.makedirs(out, exist_ok=True)
    for lang in ("en", "fa"):
        with open("/root/reference/src/sample/%s.txt" % lang, "r", encoding="utf-8") as fr, open(os.path.join(out, lang + ".txt"), "w", encoding="utf-8") as fw:
            for i, line in enumerate(fr):
                if i >= n:
                    break
                fw.write(line)
    print("sample_enfa:", {f: os.path.getsize(os.path.join(out, f)) for f in os.listdir(out)})


def make_options_kat():
    sys.path.insert(0, "/root/reference/src")
    import option_parser as ref_op
    out = {"source": "rasoolims/ImageTranslate src/option_parser.py"}
    for name in ("get_lm_option_parser", "get_img_options_parser"):
        parser = getattr(ref_op, name)()
        opts = []
        for o in parser.option_list:
            if o.dest is None:  # --help
                continue
            opts.append({"flags": o._short_opts + o._long_opts, "dest": o.dest, "type": o.type, "action": o.action,
                         "default": parser.defaults.get(o.dest)})
        out[name] = opts
    json.dump(out, open(os.path.join(HERE, "options_kat.json"), "w"), indent=1)
    print("options_kat:", {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    make_config_kat()
    make_loss_kat()
    make_toy()
    make_beam_kat()
    make_beam()
    make_beam_step1_kat()
    make_marshal_kat()
    make_options_kat()
    make_sample_excerpt()
    print("wrote", os.listdir(HERE))
