"""Translate a text file with beam search -- counterpart of src/translate.py (model + tokenizer directory in, one
output line per input line, in input order)."""
from optparse import OptionParser

import torch

from .image_model import ImageMassSeq2Seq
from .seq_gen import BeamDecoder, get_outputs_until_eos
from .textprocessor import TextProcessor


def build_batches(ids_list, max_tokens: int, pad_idx: int):
    """Length-sorted source batches of at most max_tokens padded tokens; yields (original indices, ids, mask, sizes)."""
    order = sorted(range(len(ids_list)), key=lambda i: len(ids_list[i]))
    cur = []
    for i in order:
        longest = len(ids_list[i])  # sorted ascending: the candidate is the longest
        if cur and longest * (len(cur) + 1) > max_tokens:
            yield _pack(cur, ids_list, pad_idx)
            cur = []
        cur.append(i)
    if cur:
        yield _pack(cur, ids_list, pad_idx)


def _pack(idx, ids_list, pad_idx):
    width = max(len(ids_list[i]) for i in idx)
    ids = torch.full((len(idx), width), pad_idx, dtype=torch.long)
    for r, i in enumerate(idx):
        ids[r, :len(ids_list[i])] = torch.tensor(ids_list[i], dtype=torch.long)
    return idx, ids, ids != pad_idx, torch.tensor([len(ids_list[i]) for i in idx])


@torch.no_grad()
def translate_lines(model, generator, text_processor: TextProcessor, lines, src_lang: int, dst_lang: int, max_tokens: int = 4000):
    ids_list = [text_processor.tokenize_one_sentence_with_langid(ln.strip(), src_lang) for ln in lines]
    out = [""] * len(lines)
    src_l, dst_l = text_processor.languages[text_processor.id2token(src_lang)], text_processor.languages[text_processor.id2token(dst_lang)]
    for idx, ids, mask, sizes in build_batches(ids_list, max_tokens, text_processor.pad_token_id()):
        n = len(idx)
        hyp = generator(src_inputs=ids.cuda(), src_sizes=sizes, first_tokens=torch.full((n,), dst_lang, dtype=torch.long),
                        src_mask=mask.cuda(), src_langs=torch.full((n,), src_l, dtype=torch.long).cuda(),
                        tgt_langs=torch.full((n,), dst_l, dtype=torch.long).cuda(), pad_idx=text_processor.pad_token_id())
        for r, i in enumerate(idx):
            out[i] = text_processor.decode(hyp[r][1:].tolist())
    return out


def main(argv=None):
    parser = OptionParser()
    parser.add_option("--input", dest="input_path")
    parser.add_option("--output", dest="output_path")
    parser.add_option("--src", dest="src_lang", help="source language tag without brackets")
    parser.add_option("--target", dest="target_lang")
    parser.add_option("--tok", dest="tokenizer_path")
    parser.add_option("--model", dest="model_path")
    parser.add_option("--beam", dest="beam_width", type="int", default=4)
    parser.add_option("--max_len_a", dest="max_len_a", type="float", default=1.3)
    parser.add_option("--max_len_b", dest="max_len_b", type="int", default=5)
    parser.add_option("--len-penalty", dest="len_penalty_ratio", type="float", default=0.8)
    parser.add_option("--batch", dest="batch", type="int", default=4000)
    parser.add_option("--fp32", action="store_true", dest="fp32", default=False)
    options, _ = parser.parse_args(argv)
    tp = TextProcessor(options.tokenizer_path)
    model = ImageMassSeq2Seq.load(ImageMassSeq2Seq, options.model_path, tok_dir=options.tokenizer_path)
    model.set_compute_dtype(torch.float32 if options.fp32 else torch.bfloat16)
    model = model.cuda().eval()
    generator = BeamDecoder(model, beam_width=options.beam_width, max_len_a=options.max_len_a, max_len_b=options.max_len_b,
                            len_penalty_ratio=options.len_penalty_ratio)
    with open(options.input_path, "r") as fp:
        lines = [ln for ln in fp.read().split("\n") if ln.strip()]
    res = translate_lines(model, generator, tp, lines, tp.token_id("<" + options.src_lang + ">"),
                          tp.token_id("<" + options.target_lang + ">"), options.batch)
    with open(options.output_path, "w") as fw:
        fw.write("\n".join(res) + "\n")


if __name__ == "__main__":
    main()
