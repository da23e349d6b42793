"""Text-processor objects the model classes are constructed with (``text_processor`` argument of ``Seq2Seq``).

The tokenizer itself is host-side and OUT of the hot-path scope; the models only read
``tokenizer.get_vocab_size()``, the special-token ids and ``languages``.  Two implementations:

* ``SyntheticTextProcessor`` -- ids only, no tokenizer: used by the synthetic-batch benchmark and the tests.
* ``TextProcessor``          -- thin wrapper over a trained SentencePiece-BPE vocabulary directory with the same
  methods as the reference's ``src/textprocessor.py`` that the path uses (special ids follow ``:22-31``:
  pad=0, <s>=1, <unk>=2, <mask>=3, </s>=4, then the language tags).
"""
import os
import pickle
from typing import Dict, Optional


class _VocabSize:
    def __init__(self, v):
        self._v = v

    def get_vocab_size(self):
        return self._v


class SyntheticTextProcessor:
    def __init__(self, vocab_size: int = 30000, languages: Optional[Dict[str, int]] = None):
        self.languages = languages if languages is not None else {"<en>": 0, "<fa>": 1}
        self.tokenizer = _VocabSize(vocab_size)
        self.pad_token, self.bos, self.unk_token, self.mask_token, self.sep_token = "<pad>", "<s>", "<unk>", "<mask>", "</s>"
        self.special_tokens = [self.pad_token, self.bos, self.unk_token, self.mask_token, self.sep_token] + list(
            self.languages.keys())
        self.max_len = 512

    def pad_token_id(self) -> int: return 0
    def bos_token_id(self) -> int: return 1
    def unk_token_id(self) -> int: return 2
    def mask_token_id(self) -> int: return 3
    def sep_token_id(self) -> int: return 4
    def vocab_size(self) -> int: return self.tokenizer.get_vocab_size()

    def token_id(self, token: str) -> int:
        return self.special_tokens.index(token) if token in self.special_tokens else 0

    def lang_id(self, tok) -> int:
        return self.languages.get(tok, 0)

    def id2token(self, id: int) -> str:
        return self.special_tokens[int(id)] if 0 <= int(id) < len(self.special_tokens) else "tok%d" % int(id)

    def is_lang(self, id) -> bool:
        return 5 <= int(id) < 5 + len(self.languages)


class TextProcessor:
    def __init__(self, tok_model_path: Optional[str] = None):
        self.languages: Dict[str, int] = {}
        self.tokenizer = None
        if tok_model_path is not None:
            full = os.path.join(tok_model_path, "tokenizer.json")
            if os.path.exists(full):  # written by train_tokenizer below: the complete tokenizer in one file
                from tokenizers import Tokenizer
                self.tokenizer = Tokenizer.from_file(full)
            else:  # a directory written by the reference (vocab.json + merges.txt)
                from tokenizers import SentencePieceBPETokenizer
                self.tokenizer = SentencePieceBPETokenizer(os.path.join(tok_model_path, "vocab.json"),
                                                           os.path.join(tok_model_path, "merges.txt"))
            from .safe_pickle import load_langs  # the reference's pickle file, read without executing anything from it
            self.languages = load_langs(os.path.join(tok_model_path, "langs"))
        self._init_properties(self.languages)

    def _init_properties(self, languages):
        self.max_len = 512
        self.pad_token, self.bos, self.unk_token, self.mask_token, self.sep_token = "<pad>", "<s>", "<unk>", "<mask>", "</s>"
        self.special_tokens = [self.pad_token, self.bos, self.unk_token, self.mask_token, self.sep_token] + list(
            languages.keys())
        self.languages = languages

    def train_tokenizer(self, paths, vocab_size: int, to_save_dir: str, languages: Dict[str, int]):
        from tokenizers import SentencePieceBPETokenizer
        self.tokenizer = SentencePieceBPETokenizer()
        self._init_properties(languages)
        def lines():
            for path in paths:
                with open(path, "r") as fp:
                    for line in fp:
                        if line.strip():
                            yield line.strip()  # no trailing newline: it would be learnt as a symbol
        self.tokenizer.train_from_iterator(lines(), vocab_size=vocab_size, min_frequency=5, special_tokens=self.special_tokens)
        os.makedirs(to_save_dir, exist_ok=True)
        self.tokenizer.save_model(to_save_dir)  # vocab.json + merges.txt (the reference's layout)
        self.tokenizer.save(os.path.join(to_save_dir, "tokenizer.json"))
        with open(os.path.join(to_save_dir, "langs"), "wb") as fp:
            pickle.dump(self.languages, fp)

    def pad_token_id(self) -> int: return self.tokenizer.token_to_id(self.pad_token)
    def mask_token_id(self) -> int: return self.tokenizer.token_to_id(self.mask_token)
    def unk_token_id(self) -> int: return self.tokenizer.token_to_id(self.unk_token)
    def bos_token_id(self) -> int: return self.tokenizer.token_to_id(self.bos)
    def sep_token_id(self) -> int: return self.tokenizer.token_to_id(self.sep_token)
    def vocab_size(self) -> int: return self.tokenizer.get_vocab_size()

    def token_id(self, token: str) -> int:
        tok_id = self.tokenizer.token_to_id(token)
        return 0 if tok_id is None else tok_id

    def id2token(self, id: int) -> str:
        return self.tokenizer.id_to_token(id)

    def is_lang(self, id) -> bool:
        return self.tokenizer.id_to_token(int(id)) in self.languages

    def lang_id(self, tok) -> int:
        return self.languages.get(tok, 0)

    def tokenize_one_sentence_with_langid(self, line, lang_id):
        """ids of ``line`` framed as [lang_id] ... </s> (src/textprocessor.py:74-76), truncated to 512."""
        return ([lang_id] + self.tokenizer.encode(line).ids + [self.token_id(self.sep_token)])[:512]

    def tokenize(self, lines):
        rows = [ln.strip() for ln in lines.strip().split("\n") if ln.strip()]
        return [enc.ids for enc in self.tokenizer.encode_batch(rows)]

    def decode(self, ids) -> str:
        """ids -> text without the special tokens (language tag, </s>, padding)."""
        specials = {self.token_id(t) for t in self.special_tokens}
        return self.tokenizer.decode([int(i) for i in ids if int(i) not in specials], skip_special_tokens=True)

    def tokenize_one_sentence(self, line):
        """'<lang> words ... </s>' -> ids (language tag first, end-of-sentence last), truncated to 512."""
        parts = line.strip().split(" ")
        ids = [self.token_id(parts[0])] + self.tokenizer.encode(" ".join(parts[1:-1])).ids + [self.token_id(parts[-1])]
        return ids[:512]
