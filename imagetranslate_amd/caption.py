"""Caption a directory of images with beam search -- counterpart of src/caption.py (``build_model`` ``:62-73``,
``build_data_loader`` ``:49-59``, ``caption_batch`` ``:32-46``; same flags).  The image directory holds pre-extracted
region features (``features.pt``, dataset.RegionFeatures) in place of pixels: the CNN trunk is outside the hot path."""
import datetime
from optparse import OptionParser

import torch

from . import dataset
from .image_model import ImageCaptioning
from .seq2seq import Seq2Seq
from .seq_gen import BeamDecoder


def get_lm_option_parser():
    parser = OptionParser()
    for flag, dest, kind, default in (("--input", "input_path", "string", None), ("--target", "target_lang", "string", None),
                                      ("--output", "output_path", "string", None), ("--batch", "batch", "int", 16),
                                      ("--tok", "tokenizer_path", "string", None), ("--model", "model_path", "string", None),
                                      ("--beam", "beam_width", "int", 4), ("--max_len_a", "max_len_a", "float", 1.3),
                                      ("--max_len_b", "max_len_b", "int", 5), ("--len-penalty", "len_penalty_ratio", "float", 0.8),
                                      ("--max-len", "max_len", "int", 256)):
        parser.add_option(flag, dest=dest, type=kind, default=default)
    for flag, dest in (("--fp16", "fp16"), ("--obj", "obj"), ("--fp32", "fp32")):
        parser.add_option(flag, action="store_true", dest=dest, default=False)
    return parser


@torch.no_grad()
def caption_batch(batch, generator, text_processor, max_len: int = 256):
    outputs = generator(first_tokens=batch["first_tokens"], images=batch["images"], tgt_langs=batch["tgt_langs"],
                        pad_idx=text_processor.pad_token_id(), max_len=max_len)
    return [text_processor.decode(h[1:].tolist()) for h in outputs], batch["paths"]


def build_data(options, text_processor):
    assert options.target_lang is not None
    tag = "<" + options.target_lang + ">"
    return dataset.ImageDataset(options.input_path, options.batch, first_token=text_processor.token_id(tag),
                                target_lang=text_processor.languages[tag])


def build_model(options):
    model = Seq2Seq.load(ImageCaptioning, options.model_path, tok_dir=options.tokenizer_path, use_obj=options.obj)
    model.set_compute_dtype(torch.float32 if options.fp32 else torch.bfloat16)
    model = model.cuda().eval()
    generator = BeamDecoder(model, beam_width=options.beam_width, max_len_a=options.max_len_a, max_len_b=options.max_len_b,
                            len_penalty_ratio=options.len_penalty_ratio)
    return generator, model.text_processor


def main(argv=None):
    options, _ = get_lm_option_parser().parse_args(argv)
    generator, text_processor = build_model(options)
    data = build_data(options, text_processor)
    count = 0
    with open(options.output_path, "w") as writer:
        for i in range(len(data)):
            captions, paths = caption_batch(data[i], generator, text_processor, options.max_len)
            count += len(captions)
            writer.write("\n".join(p + "\t" + c for p, c in zip(paths, captions)) + "\n")
    print(datetime.datetime.now(), "Captioned", count, "images")


if __name__ == "__main__":
    main()
