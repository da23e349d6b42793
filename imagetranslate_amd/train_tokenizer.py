"""Train the BPE vocabulary directory (vocab.json, merges.txt, langs) -- counterpart of src/train_tokenizer.py.
Input text lines start with their language tag (``<en> some text </s>``); tags found become the language table."""
import os
from optparse import OptionParser

from .textprocessor import TextProcessor


def get_tokenizer(train_path=None, model_path=None, vocab_size: int = 30000) -> TextProcessor:
    if train_path is None or (model_path is not None and os.path.exists(os.path.join(model_path, "vocab.json"))):
        return TextProcessor(tok_model_path=model_path)
    languages = set()
    with open(train_path, "r") as fp:
        for line in fp:
            first = line.strip().split(" ")[0] if line.strip() else ""
            if first.startswith("<") and first.endswith(">"):
                languages.add(first)
    tp = TextProcessor()
    tp.train_tokenizer(paths=[train_path], vocab_size=vocab_size, to_save_dir=model_path,
                       languages={l: i for i, l in enumerate(sorted(languages))})
    return TextProcessor(tok_model_path=model_path)


def main(argv=None):
    parser = OptionParser()
    parser.add_option("--data", dest="data_path", help="text file, one sentence per line, language tag first")
    parser.add_option("--vocab_size", dest="vocab_size", type="int", default=30000)
    parser.add_option("--model", dest="model_path", help="output directory")
    options, _ = parser.parse_args(argv)
    tp = get_tokenizer(options.data_path, options.model_path, options.vocab_size)
    print("vocabulary size", tp.vocab_size(), "languages", tp.languages)


if __name__ == "__main__":
    main()
