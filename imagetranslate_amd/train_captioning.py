"""Captioning trainer -- counterpart of src/train_captioning.py (``ImageCaptionTrainer.train_epoch`` ``:26-141``,
``train`` ``:194-286``): image batches (region features -> ``fc`` + location embedding -> decoder) and, optionally,
MT batches as a second task whose loss is weighted by ``--mtlw`` (``:83``); gradients of both tasks accumulate, the clip
runs after every backward and the optimizer steps every ``--acc`` micro-steps (``:91-97``).  The model step is the HIP
path (``ImageCaptioning.loss_fused``).  BLEU evaluation needs sacrebleu (absent): the dev set is scored by its loss and,
with ``eval_captions``, decoded with beam search."""
import datetime
import os
import random

import torch

from . import dataset
from .image_model import ImageCaptioning, ImageMassSeq2Seq
from .option_parser import get_img_options_parser
from .parallel import clip_in_place
from .param_store import store_of
from .seq2seq import Seq2Seq
from .seq_gen import BeamDecoder, get_outputs_until_eos
from .textprocessor import TextProcessor
from .train_image_mt import ImageMTTrainer, LossMeter, init_distributed, reject_off_path
from .utils import build_optimizer


class ImageCaptionTrainer(ImageMTTrainer):
    def __init__(self, model, beam_width: int = 5, max_len_a: float = 1.3, max_len_b: int = 5, len_penalty_ratio: float = 0.8,
                 **kw):
        super().__init__(model, **kw)
        self.generator = BeamDecoder(model, beam_width=beam_width, max_len_a=max_len_a, max_len_b=max_len_b,
                                     len_penalty_ratio=len_penalty_ratio)

    # one image batch (src/train_captioning.py:40-58,76-97)
    def caption_step(self, batch, accum: int = 1):
        model, tp = self.model, self.model.text_processor
        # a batch without a single target token is decided on the host BEFORE anything is enqueued; under data parallelism
        # this rank still runs the step's collectives (on its unchanged gradient buffer) and counts the micro-step, so every
        # rank issues the same all-reduces and steps the optimizer at the same time
        if not bool(batch["caption_mask"][:, 1:].any()):
            if self.sync is not None and self.world_size > 1:
                self.sync.begin_step()
                self._finish_micro_step(None, accum, self.sync.finish())
            return 0.0, 0
        if self.sync is not None:
            self.sync.begin_step()
        loss, ntokens = model.loss_fused(tgt_inputs=batch["captions"], tgt_mask=batch["caption_mask"], pad_idx=tp.pad_token_id(),
                                         tgt_langs=batch["langs"], batch=batch)
        loss.backward()
        scale = self.sync.finish() if self.sync is not None else 1.0
        self._finish_micro_step(loss, accum, scale)
        return loss.detach(), int(ntokens)

    @torch.no_grad()
    def caption_dev_loss(self, img_dev_data):
        self.model.eval()
        tp = self.model.text_processor
        total, count = 0.0, 0
        for i in range(len(img_dev_data)):
            batch = img_dev_data[i]
            loss, n = self.model.loss_fused(tgt_inputs=batch["captions"], tgt_mask=batch["caption_mask"], pad_idx=tp.pad_token_id(),
                                            tgt_langs=batch["langs"], batch=batch)
            total += float(loss) * int(n)
            count += int(n)
        self.model.train()
        return total / max(count, 1)

    @torch.no_grad()
    def eval_captions(self, img_test_data, max_batches: int = None):
        """Beam-search captions of a test dataset (src/train_captioning.py:143-165, without the sacrebleu scoring):
        {image id: caption text}."""
        model, tp = self.model, self.model.text_processor
        model.eval()
        out = {}
        for i in range(len(img_test_data) if max_batches is None else min(max_batches, len(img_test_data))):
            b = img_test_data[i]
            hyps = self.generator(images=b["images"], first_tokens=b["first_tokens"], tgt_langs=b["langs"],
                                  pad_idx=tp.pad_token_id(), max_len=b["max_len"])
            for image_id, h in zip(b["img_ids"], hyps):
                out[image_id] = tp.decode(h[1:].tolist()) if hasattr(tp, "decode") else h[1:].tolist()
        model.train()
        return out

    def train_epoch(self, img_data=None, mt_data=None, img_dev_data=None, mt_dev_data=None, step: int = 0,
                    max_step: int = 300000, save_path: str = None, accum: int = 1, mtl_weight: float = 0.1,
                    log_every: int = 50, eval_every: int = 5000, **kwargs):
        order = [("img", i) for i in range(len(img_data) if img_data is not None else 0)]
        order += [("mt", i) for i in range(len(mt_data or []))]
        random.Random(self.seed + self.epoch).shuffle(order)
        if self.world_size > 1 and order:
            order = order + order[:(-len(order)) % self.world_size]
        order = order[self.rank::self.world_size]
        self.epoch += 1
        meter, t0 = LossMeter(), datetime.datetime.now()
        for kind, i in order:
            if step >= max_step:
                break
            try:
                if kind == "img":
                    loss, n = self.caption_step(img_data[i], accum)
                else:  # MT data as the second task (:59-75), weighted by mtl_weight (:83)
                    loss, n = self.mt_step(mt_data[i], accum, loss_weight=mtl_weight)
            except RuntimeError as err:
                if self.world_size > 1:
                    raise
                print("skipping batch:", repr(err))
                self.optimizer.zero_grad()
                continue
            if n == 0 and self.world_size == 1:
                continue
            step += 1  # (an empty batch of a multi-rank job counts: the other ranks counted theirs)
            meter.add(loss, n)
            if step % log_every == 0:
                mean, tokens = meter.read()
                if self.rank == 0:
                    secs = (datetime.datetime.now() - t0).total_seconds()
                    print(datetime.datetime.now(), "Epoch Step: %d Loss: %f Tokens per Sec: %f " % (step, mean, tokens / max(secs, 1e-9)), flush=True)
                t0 = datetime.datetime.now()
            if step % eval_every == 0:
                self._validate(img_dev_data, mt_dev_data, save_path)
        return step

    def _validate(self, img_dev_data, mt_dev_data, save_path):
        score = None
        if img_dev_data is not None:
            score = self.caption_dev_loss(img_dev_data)
            if self.rank == 0:
                print(datetime.datetime.now(), "caption dev loss %.4f (best %.4f)" % (score, self.best_loss), flush=True)
        if mt_dev_data is not None:
            mt = self.dev_loss(mt_dev_data)
            if self.rank == 0:
                print(datetime.datetime.now(), "MT dev loss %.4f" % mt, flush=True)
            score = mt if score is None else score
        if self.rank == 0 and save_path:
            self.model.save(save_path + ".latest")
            if score is not None and score < self.best_loss:
                self.model.save(save_path)
        if score is not None:
            self.best_loss = min(self.best_loss, score)

    @staticmethod
    def train(options):
        reject_off_path(options, lm_supported=True)
        rank, world = init_distributed()
        random.seed(options.seed)
        torch.manual_seed(options.seed)
        if options.model_path and not os.path.exists(options.model_path):
            os.makedirs(options.model_path, exist_ok=True)
        tp = TextProcessor(options.tokenizer_path)
        assert tp.pad_token_id() == 0
        if options.pretrained_path is not None:
            caption_model = Seq2Seq.load(ImageCaptioning, options.pretrained_path, tok_dir=options.tokenizer_path,
                                         use_obj=not options.no_obj)
        else:
            caption_model = ImageCaptioning(text_processor=tp, tie_embed=options.tie_embed, resnet_depth=options.resnet_depth,
                                            lang_dec=options.lang_decoder, enc_layer=options.encoder_layer,
                                            dec_layer=options.decoder_layer, embed_dim=options.embed_dim,
                                            intermediate_dim=options.intermediate_layer_dim, use_obj=not options.no_obj,
                                            num_attention_heads=options.heads, image_feat_dim=options.feat_dim)
        if options.lm_path is not None:  # in the reference this is a pretrained MT model whose stacks are adopted (:212-220)
            mt_model = Seq2Seq.load(ImageMassSeq2Seq, options.lm_path, tok_dir=options.tokenizer_path)
            assert len(caption_model.encoder.encoder.layer) == len(mt_model.encoder.encoder.layer)
            assert len(caption_model.decoder.decoder.layer) == len(mt_model.decoder.decoder.layer)
            caption_model.encoder = mt_model.encoder
            caption_model.decoder = mt_model.decoder
            caption_model.output_layer = mt_model.output_layer
        caption_model.set_compute_dtype(torch.float32 if options.fp32 else torch.bfloat16)
        caption_model = caption_model.cuda().train()
        optimizer = build_optimizer(caption_model, options.learning_rate, options.warmup)
        trainer = ImageCaptionTrainer(model=caption_model, mask_prob=options.mask_prob, optimizer=optimizer, clip=options.clip,
                                      beam_width=options.beam_width, max_len_a=options.max_len_a, max_len_b=options.max_len_b,
                                      len_penalty_ratio=options.len_penalty_ratio, rank=rank, world_size=world, seed=options.seed)
        feats = dataset.RegionFeatures(options.image_dir)
        mk_img = lambda cls, path: cls(root_img_dir=options.image_dir, data_bin_file=path, max_capacity=options.img_capacity,
                                       text_processor=tp, max_img_per_batch=options.max_image, features=feats)
        img_train = mk_img(dataset.ImageCaptionDataset, options.train_path)
        img_dev = mk_img(dataset.ImageCaptionDataset, options.dev_path) if options.dev_path else None
        pad = tp.pad_token_id()
        mk_mt = lambda path: dataset.MTDataset(max_batch_capacity=options.total_capacity, max_batch=options.batch, pad_idx=pad,
                                               max_seq_len=options.max_seq_len, batch_pickle_dir=path).batches
        mt_train = sum((mk_mt(p.strip()) for p in (options.mt_train_path or "").split(",") if p.strip()), []) or None
        mt_dev = sum((mk_mt(p.strip()) for p in (options.mt_dev_path or "").split(",") if p.strip()), []) or None
        if rank == 0:
            print("image batches", len(img_train), "MT batches", len(mt_train or []), flush=True)
        step, epoch = 0, 1
        while options.step > 0 and step < options.step and epoch <= options.num_epochs:
            step = trainer.train_epoch(img_data=img_train, mt_data=mt_train, img_dev_data=img_dev, mt_dev_data=mt_dev, step=step,
                                       max_step=options.step, save_path=options.model_path, accum=options.accum,
                                       mtl_weight=options.mtl_weight, log_every=options.log_steps, eval_every=options.eval_steps)
            epoch += 1
        trainer._validate(img_dev, mt_dev, options.model_path)
        if rank == 0 and options.model_path and not os.path.exists(os.path.join(options.model_path, "mt_model.state_dict")):
            caption_model.save(options.model_path)
        return trainer


def main(argv=None):
    options, _ = get_img_options_parser().parse_args(argv)
    ImageCaptionTrainer.train(options)
    print("Finished Training!")


if __name__ == "__main__":
    main()
