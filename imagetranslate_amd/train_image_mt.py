"""Training entry point for the text path -- thin counterpart of src/train_image_mt.py (``ImageMTTrainer``
``:44-333`` and ``train`` ``:394-560``): supervised MT batches and/or MASS batches, label-smoothed loss, gradient
clipping + inverse-sqrt Adam, dev-set loss, best-checkpoint saving, one process per GPU under torch.distributed.
Round 3 adds the back-translation phase (``--fstep`` / ``--langs`` / ``--bt-beam``, src/train_image_mt.py:108-198,509-533): after
``--step`` ordinary steps the optimizer schedule is reset and every monolingual batch is translated by the model itself (KV-cached
beam search, no gradient) and trained on as (translation -> original).  What the reference trainer does around the step that needs
absent packages (apex, sacrebleu, the torchvision image pipeline) is left out; the model step itself is the HIP path."""
import datetime
import math
import os
import random
import torch

from torch.nn.utils.rnn import pad_sequence

from . import dataset
from .image_model import ImageMassSeq2Seq
from .option_parser import get_img_options_parser
from .parallel import GradSync, clip_in_place, train_step
from .textprocessor import TextProcessor
from .utils import build_optimizer, mass_mask_device


class ImageMTTrainer:
    def __init__(self, model, mask_prob: float = 0.3, clip: float = 1.0, optimizer=None, rank: int = 0, world_size: int = 1,
                 seed: int = 1234, **kwargs):
        self.model = model
        self.clip = clip
        self.optimizer = optimizer
        self.mask_prob = mask_prob
        self.rank, self.world_size = rank, world_size
        self.sync = GradSync(model) if world_size > 1 else None
        self.best_loss = float("inf")
        self.seed = seed
        self.epoch = 0
        self.micro_step = 0  # backward passes so far: the optimizer steps when this reaches a multiple of `accum`
        # MASS span / replacement draws: a generator of this rank's own, so that the ranks' batch ORDER (drawn from a
        # generator every rank seeds identically, below) never depends on how many MASS batches a rank has seen
        self._mass_rng = random.Random((seed + 1) * 7919 + rank)
        self.generator = None  # BeamDecoder of the back-translation phase (built on first use)
        self.bt_kw = dict(beam_width=kwargs.get("bt_beam_width", 1), max_len_a=kwargs.get("max_len_a", 1.3),
                          max_len_b=kwargs.get("max_len_b", 5), len_penalty_ratio=kwargs.get("len_penalty_ratio", 0.8))

    def _finish_micro_step(self, loss, accum: int, scale: float):
        """clip after EVERY backward, step every `accum` (src/train_image_mt.py:291-295)."""
        self.micro_step += 1
        if self.micro_step % max(1, accum) == 0:
            self.optimizer.step(max_grad_norm=self.clip, grad_scale=scale, zero_grad=True)
        else:
            from .param_store import store_of
            clip_in_place(self.optimizer, store_of(self.model.encoder).ensure(), self.clip, scale)

    # one MT batch (src/train_image_mt.py:239-295)
    def mt_step(self, batch, accum: int = 1, loss_weight: float = 1.0):
        batch = {k: (v[0] if isinstance(v, list) else v) for k, v in batch.items()}
        loss, ntokens = train_step(self.model, self.optimizer, batch, sync=self.sync, clip=self.clip,
                                   update=((self.micro_step + 1) % max(1, accum) == 0), loss_weight=loss_weight)
        self.micro_step += 1  # only a micro-step whose backward ran counts (a failing batch is skipped, train_epoch)
        return loss.detach(), int(ntokens)  # the loss stays on the device: reading it here would stall the host every step

    # one MASS batch (src/train_image_mt.py:186-236): mask a span, recover it with its original positions
    def mass_step(self, batch, accum: int = 1):
        tp = self.model.text_processor
        src = batch["src_texts"].cuda()  # a device copy: the dataset's tensor is never modified, no unmask needed
        masked = mass_mask_device(self.mask_prob, batch["pad_idx"], src, tp, seed=self._mass_rng.getrandbits(62))
        if self.sync is not None:
            self.sync.begin_step()
        loss, ntokens = self.model.loss_fused(src_inputs=masked["src_text"], tgt_inputs=masked["to_recover"],
                                              src_langs=batch["langs"], pad_idx=tp.pad_token_id(),
                                              tgt_positions=masked["positions"])
        loss.backward()
        scale = self.sync.finish() if self.sync is not None else 1.0
        self._finish_micro_step(loss, accum, scale)
        return loss.detach(), int(ntokens)

    @staticmethod
    def get_lang_dirs(bt_langs: str, text_processor):
        """``--langs en,fa`` -> {token id of <en>: token id of <fa>, and back} (src/train_image_mt.py:535-548); None unless two
        languages are given."""
        langs = {text_processor.token_id("<" + l.strip() + ">") for l in (bt_langs or "").strip().split(",") if l.strip()}
        if len(langs) < 2:
            return None
        assert len(langs) == 2, "back-translation is defined for one language pair (src/train_image_mt.py:541)"
        a, b = sorted(langs)
        return {a: b, b: a}

    # one monolingual batch of the back-translation phase (src/train_image_mt.py:108-198, the is_mass_batch branch): translate the
    # sentences into the other language with the current model (eval mode, no gradient: "we do not backpropagate the data
    # generator following the MASS paper"), then train on (translation -> original)
    def bt_step(self, batch, lang_directions, accum: int = 1):
        from .seq_gen import BeamDecoder
        model, tp = self.model, self.model.text_processor
        pad = tp.pad_token_id()
        src = batch["src_texts"]
        src_mask = src != pad
        first = [int(l) for l in src[:, 0]]
        target_tags = torch.LongTensor([lang_directions[l] for l in first])               # first token of each translation
        dst_langs = torch.LongTensor([tp.languages[tp.id2token(lang_directions[l])] for l in first])
        if self.generator is None:
            self.generator = BeamDecoder(model, **self.bt_kw)
        was_training = model.training
        model.eval()
        with torch.no_grad():
            outs = self.generator(src_inputs=src, src_sizes=batch["pad_idx"], first_tokens=target_tags, src_langs=batch["langs"],
                                  tgt_langs=dst_langs, pad_idx=pad, src_mask=src_mask, unpad_output=False,
                                  beam_width=self.bt_kw["beam_width"])
        model.train(was_training)
        translations = pad_sequence([o.cpu() for o in outs], batch_first=True, padding_value=pad)
        if self.sync is not None:
            self.sync.begin_step()
        loss, ntokens = model.loss_fused(src_inputs=translations, tgt_inputs=src, src_langs=dst_langs, tgt_langs=batch["langs"],
                                         pad_idx=pad)
        loss.backward()
        scale = self.sync.finish() if self.sync is not None else 1.0
        self._finish_micro_step(loss, accum, scale)
        return loss.detach(), int(ntokens)

    @torch.no_grad()
    def dev_loss(self, dev_data):
        self.model.eval()
        total, count = 0.0, 0
        for batch in dev_data:
            loss, n = self.model.loss_fused(src_inputs=batch["src_texts"], tgt_inputs=batch["dst_texts"],
                                            src_langs=batch["src_langs"], tgt_langs=batch["dst_langs"])
            total += float(loss) * int(n)
            count += int(n)
        self.model.train()
        return total / max(count, 1)

    def epoch_order(self, n_mt: int, n_mass: int):
        """This rank's share of the epoch's batches: the same shuffle on every rank (a generator seeded with seed + epoch,
        nothing else draws from it), padded by wrapping around to a multiple of the world size like torch's
        DistributedSampler (src/train_image_mt.py:586-589), then strided -- every rank runs the SAME number of steps, so
        no rank is left waiting in an all-reduce when an epoch ends."""
        order = [("mt", i) for i in range(n_mt)] + [("mass", i) for i in range(n_mass)]
        random.Random(self.seed + self.epoch).shuffle(order)
        if self.world_size > 1 and order:
            short = (-len(order)) % self.world_size
            order = order + order[:short]
        return order[self.rank::self.world_size]

    def train_epoch(self, mt_data=None, mass_data=None, dev_data=None, step: int = 0, max_step: int = 10 ** 9,
                    save_path: str = None, log_every: int = 50, eval_every: int = 500, accum: int = 1, fine_tune: bool = False,
                    lang_directions=None):
        order = self.epoch_order(len(mt_data or []), len(mass_data or []))
        self.epoch += 1
        meter, t0 = LossMeter(), datetime.datetime.now()
        for kind, i in order:
            if step >= max_step:
                break
            try:
                if kind == "mt":
                    loss, n = self.mt_step(mt_data[i], accum)
                elif fine_tune:  # back-translation phase: the monolingual batches are translated and trained on (:108-198)
                    loss, n = self.bt_step(mass_data[i], lang_directions, accum)
                else:
                    loss, n = self.mass_step(mass_data[i], accum)
            except RuntimeError as err:  # the reference trainer skips a failing batch and goes on (:327-333)
                if self.world_size > 1:
                    raise  # the other ranks are inside this step's collectives: skipping here would leave them hanging
                print("skipping batch:", repr(err))
                self.optimizer.zero_grad()
                continue
            step += 1
            meter.add(loss, n)
            if step % log_every == 0:
                mean, tokens = meter.read()  # the only host read of the losses: once per log_every steps, on every rank
                if self.rank == 0:
                    secs = (datetime.datetime.now() - t0).total_seconds()
                    print(datetime.datetime.now(), "step", step, "loss %.4f" % mean, "tokens/s %.0f" % (tokens / max(secs, 1e-9)),
                          "lr %.2e" % self.optimizer.param_groups[0]["lr"], flush=True)
                t0 = datetime.datetime.now()
            if dev_data is not None and step % eval_every == 0:
                self.validate_and_save(dev_data, save_path)
        return step

    def validate_and_save(self, dev_data, save_path):
        dl = self.dev_loss(dev_data)
        if self.rank == 0:
            print(datetime.datetime.now(), "dev loss %.4f (best %.4f)" % (dl, self.best_loss), flush=True)
            if dl < self.best_loss and save_path:
                self.model.save(save_path)
        self.best_loss = min(self.best_loss, dl)
        return dl


class LossMeter:
    """Token-weighted running mean of per-step losses that stay on the device until ``read()`` (the reference reads
    ``loss.item()`` every step, src/train_image_mt.py:284: a host stall per step that lets the GPU idle while the host
    prepares the next batch)."""

    def __init__(self):
        self.items = []

    def add(self, loss, n: int):
        if n:
            self.items.append((loss, int(n)))

    def read(self):
        tokens = sum(n for _, n in self.items)
        total = sum(float(l) * n for l, n in self.items)
        self.items = []
        return total / max(tokens, 1), tokens


def get_option_parser():
    """The reference's flags (src/option_parser.py:37-88: --train_mt, --dev_mt, --mass_train, --acc, --fp16, --beam ...)
    plus the build's additions (option_parser.py)."""
    return get_img_options_parser()


def init_distributed():
    """One process per GPU (src/utils.py:93-97, src/train_image_mt.py:72-76); backend "nccl" is RCCL on ROCm."""
    rank, world = 0, 1
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("nccl")
        rank, world = dist.get_rank(), dist.get_world_size()
    return rank, world


def reject_off_path(options, lm_supported: bool = False):
    """Flags whose feature is not built are refused, never silently ignored: a requested pretrained initialisation or a
    resumed optimizer that quietly trains from scratch is worse than an error."""
    for flag, val in (("--dict", options.dict_path),):
        if val:
            raise NotImplementedError("%s: outside the hot path (lexical proposals, DESIGN section 7)" % flag)
    for flag, val in (("--lm", None if lm_supported else options.lm_path), ("--cont", options.continue_train), ("--save-opt", options.save_opt)):
        if val:
            raise NotImplementedError("%s: not implemented by this trainer (masked-LM initialisation / pickled-optimizer resume, "
                                      "src/train_image_mt.py:319-321,449-462); use --pretrained to continue from saved weights" % flag)


def train(options):
    reject_off_path(options)
    rank, world = init_distributed()
    random.seed(options.seed)
    torch.manual_seed(options.seed)
    tp = TextProcessor(options.tokenizer_path)
    if options.pretrained_path:
        model = ImageMassSeq2Seq.load(ImageMassSeq2Seq, options.pretrained_path, tok_dir=options.tokenizer_path)
    else:
        model = ImageMassSeq2Seq(text_processor=tp, lang_dec=options.lang_decoder, tie_embed=options.tie_embed,
                                 enc_layer=options.encoder_layer, dec_layer=options.decoder_layer, embed_dim=options.embed_dim,
                                 intermediate_dim=options.intermediate_layer_dim, num_attention_heads=options.heads,
                                 resnet_depth=options.resnet_depth, image_feat_dim=options.feat_dim)
    model.set_compute_dtype(torch.float32 if options.fp32 else torch.bfloat16)
    model = model.cuda().train()
    optimizer = build_optimizer(model, options.learning_rate, options.warmup)
    # (GradSync broadcasts rank 0's parameters, like the DDP constructor at src/train_image_mt.py:73)
    trainer = ImageMTTrainer(model, mask_prob=options.mask_prob, clip=options.clip, optimizer=optimizer, rank=rank, world_size=world,
                             seed=options.seed, bt_beam_width=options.bt_beam_width, max_len_a=options.max_len_a,
                             max_len_b=options.max_len_b, len_penalty_ratio=options.len_penalty_ratio)
    pad = tp.pad_token_id()
    mk = lambda cls, path, **kw: cls(max_batch_capacity=options.total_capacity, max_batch=options.batch, pad_idx=pad,
                                     max_seq_len=options.max_seq_len, ngpu=1, **kw, **path)
    mt_train, mass_train, mt_dev = [], [], None
    for pth in (options.mt_train_path or "").split(","):
        if pth.strip():
            mt_train += mk(dataset.MTDataset, dict(batch_pickle_dir=pth.strip())).batches
    for pth in (options.mass_train_path or "").split(","):
        if pth.strip():
            mass_train += mk(dataset.MassDataset, dict(batch_pickle_dir=pth.strip())).batches
    for pth in (options.mt_dev_path or "").split(","):
        if pth.strip():
            mt_dev = (mt_dev or []) + mk(dataset.MTDataset, dict(batch_pickle_dir=pth.strip())).batches
    if rank == 0:
        print("MT batches", len(mt_train), "MASS batches", len(mass_train), "dev batches", len(mt_dev or []), flush=True)
    step = 0
    for epoch in range(options.num_epochs):
        if step >= options.step:
            break
        step = trainer.train_epoch(mt_data=mt_train, mass_data=mass_train, dev_data=mt_dev, step=step, max_step=options.step,
                                   save_path=options.model_path, log_every=options.log_steps, eval_every=options.eval_steps,
                                   accum=options.accum)
    # back-translation phase (src/train_image_mt.py:509-533): optimizer schedule reset, then --fstep more steps in which the
    # monolingual batches are back-translated.  The reference's flag default is 125000 steps; here the phase runs only when a
    # language pair is given (--langs), which is what makes it well-defined.
    lang_directions = ImageMTTrainer.get_lang_dirs(options.bt_langs, tp)
    if lang_directions is not None and options.finetune_step > 0 and mass_train:
        optimizer.reset()
        total = step + options.finetune_step
        for epoch in range(options.num_epochs):
            if step >= total:
                break
            step = trainer.train_epoch(mt_data=None if options.ignore_mt_mass else mt_train, mass_data=mass_train, dev_data=mt_dev,
                                       step=step, max_step=total, save_path=options.model_path, log_every=options.log_steps,
                                       eval_every=options.eval_steps, accum=options.accum, fine_tune=True,
                                       lang_directions=lang_directions)
    if mt_dev is not None:
        trainer.validate_and_save(mt_dev, options.model_path)
    elif rank == 0 and options.model_path:
        model.save(options.model_path)
    return trainer


def main(argv=None):
    options, _ = get_option_parser().parse_args(argv)
    train(options)


if __name__ == "__main__":
    main()
