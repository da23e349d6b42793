"""Tokenise text files into the marshal example files the datasets read -- counterpart of src/create_mt_batches.py
(same file format: list of (src_ids, dst_ids, src_lang, dst_lang) sorted by target length, or of (src_ids, lang)
sorted by length, written as ``<output>.<part>`` for monolingual data)."""
import marshal
from optparse import OptionParser

from .textprocessor import TextProcessor


def write(text_processor: TextProcessor, output_file: str, src_txt_file: str, src_lang: int, dst_txt_file: str = None,
          dst_lang: int = None, min_len: int = 1, max_len: int = 175, part_size: int = 6000000):
    src_lang_idx = text_processor.languages[text_processor.id2token(src_lang)]
    if dst_txt_file is not None:
        dst_lang_idx = text_processor.languages[text_processor.id2token(dst_lang)]
        examples = []
        with open(src_txt_file, "r") as s_fp, open(dst_txt_file, "r") as d_fp:
            for src_line, dst_line in zip(s_fp, d_fp):
                src_line, dst_line = src_line.strip(), dst_line.strip()
                if not src_line or not dst_line:
                    continue
                s = text_processor.tokenize_one_sentence_with_langid(src_line, src_lang)
                d = text_processor.tokenize_one_sentence_with_langid(dst_line, dst_lang)
                if min_len <= len(s) <= max_len and min_len <= len(d) <= max_len:
                    examples.append((s, d, src_lang_idx, dst_lang_idx))
        examples.sort(key=lambda e: len(e[1]))  # stable: file order among equal target lengths
        with open(output_file, "wb") as fw:
            marshal.dump(examples, fw)
        return len(examples)
    total, part, examples = 0, 0, []

    def flush():
        nonlocal part, examples
        examples.sort(key=lambda e: len(e[0]))
        with open(output_file + "." + str(part), "wb") as fw:
            marshal.dump(examples, fw)
        part, examples = part + 1, []

    with open(src_txt_file, "r") as s_fp:
        for src_line in s_fp:
            src_line = src_line.strip()
            if not src_line:
                continue
            s = text_processor.tokenize_one_sentence_with_langid(src_line, src_lang)
            if min_len <= len(s) <= max_len:
                examples.append((s, src_lang_idx))
                total += 1
            if len(examples) >= part_size:
                flush()
    if examples:
        flush()
    return total


def main(argv=None):
    parser = OptionParser()
    parser.add_option("--src", dest="src_data_path")
    parser.add_option("--dst", dest="dst_data_path", default=None)
    parser.add_option("--output", dest="output_path")
    parser.add_option("--tok", dest="tokenizer_path")
    parser.add_option("--src-lang", dest="src_lang", help="source language tag without brackets, e.g. en")
    parser.add_option("--dst-lang", dest="dst_lang", default=None)
    parser.add_option("--min_seq_len", dest="min_seq_len", type="int", default=1)
    parser.add_option("--max_seq_len", dest="max_seq_len", type="int", default=175)
    options, _ = parser.parse_args(argv)
    tp = TextProcessor(options.tokenizer_path)
    src_lang = tp.token_id("<" + options.src_lang + ">")
    dst_lang = tp.token_id("<" + options.dst_lang + ">") if options.dst_lang else None
    n = write(tp, options.output_path, options.src_data_path, src_lang, options.dst_data_path, dst_lang, options.min_seq_len,
              options.max_seq_len)
    print("wrote", n, "examples to", options.output_path)


if __name__ == "__main__":
    main()
