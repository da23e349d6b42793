"""Command-line flags of the trainers -- the names, destinations and defaults of the reference's
``src/option_parser.py:4-88`` (``get_lm_option_parser`` + ``get_img_options_parser``), so that the reference's command
lines (README.md:160-163, :212-216) parse here unchanged.  Kept as a table: (flag, dest, kind, default).

Flags the reference parses but never reads on this path (``--dropout``, ``--nll``, ``--dff``, ``--max_grad_norm``,
``--cache_size`` ...) are accepted and ignored; flags whose feature is outside the hot path
(``--dict``, ``--lm`` for train_image_mt, ``--cont`` / ``--save-opt``) are accepted by the parser and refused by the
trainers (``reject_off_path``), never silently ignored; ``--langs`` / ``--fstep`` / ``--bt-beam`` drive the back-translation phase.  Build additions, at the end of the table: ``--heads`` (the reference hard-codes 12,
src/lm_config.py:13), ``--fp32`` (compute in fp32; default is bf16 whether or not ``--fp16`` is given: apex fp16 maps to
bf16 MFMA on MI355X), ``--seed``, ``--eval-steps`` / ``--log-steps`` / ``--save-steps`` (the reference hard-codes 5000 / 50 /
10000, src/train_image_mt.py:302-325), ``--feat-dim`` (channels of the pre-extracted region features).
"""
from optparse import OptionParser

_LM = [
    ("--train", "train_path", "str", None), ("--dev", "dev_path", "str", None), ("--tok", "tokenizer_path", "str", None),
    ("--cache_size", "cache_size", "int", 300), ("--model", "model_path", "str", None),
    ("--pretrained", "pretrained_path", "str", None), ("--epoch", "num_epochs", "int", 100), ("--clip", "clip", "int", 1),
    ("--batch", "batch", "int", 6000), ("--mask", "mask_prob", "float", 0.15), ("--lr", "learning_rate", "float", 0.0001),
    ("--warmup", "warmup", "int", 12500), ("--step", "step", "int", 125000), ("--max_grad_norm", "max_grad_norm", "float", 1.0),
    ("--cont", "continue_train", "flag", False), ("--dropout", "dropout", "float", 0.1), ("--dff", "d_ff", "int", 2048),
    ("--reformer", "reformer", "flag", False), ("--enc", "encoder_layer", "int", 6), ("--embed", "embed_dim", "int", 768),
    ("--intermediate", "intermediate_layer_dim", "int", 3072), ("--local_rank", "local_rank", "int", -1),
]

_IMG = [
    ("--capacity", "total_capacity", "int", 600), ("--lm", "lm_path", "str", None), ("--dict", "dict_path", "str", None),
    ("--beam", "beam_width", "int", 5), ("--bt-beam", "bt_beam_width", "int", 1), ("--max_len_a", "max_len_a", "float", 1.3),
    ("--max_len_b", "max_len_b", "int", 5), ("--len-penalty", "len_penalty_ratio", "float", 0.8),
    ("--max_seq_len", "max_seq_len", "int", 175), ("--ldec", "lang_decoder", "flag", False), ("--nll", "nll_loss", "flag", False),
    ("--fp16", "fp16", "flag", False), ("--dev_mt", "mt_dev_path", "str", None), ("--train_mt", "mt_train_path", "str", None),
    ("--fstep", "finetune_step", "int", 125000), ("--mass_train", "mass_train_path", "str", None), ("--image", "image_dir", "str", ""),
    ("--img_capacity", "img_capacity", "int", 50), ("--max-image", "max_image", "int", 32), ("--img-depth", "resnet_depth", "int", 1),
    ("--langs", "bt_langs", "str", ""), ("--mmode", "mm_mode", "str", "mixed"), ("--dec", "decoder_layer", "int", 6),
    ("--ignore-mt-mass", "ignore_mt_mass", "flag", False), ("--tie", "tie_embed", "flag", False), ("--output", "output", "str", None),
    ("--src-neg", "src_neg", "str", None), ("--dst-neg", "dst_neg", "str", None), ("--save-opt", "save_opt", "flag", False),
    ("--no-obj", "no_obj", "flag", False), ("--acc", "accum", "int", 1), ("--mtlw", "mtl_weight", "float", 0.1),
]

_BUILD = [
    ("--heads", "heads", "int", 12), ("--fp32", "fp32", "flag", False), ("--seed", "seed", "int", 1234),
    ("--eval-steps", "eval_steps", "int", 5000), ("--log-steps", "log_steps", "int", 50), ("--save-steps", "save_steps", "int", 10000),
    ("--feat-dim", "feat_dim", "int", None),
]


def _add(parser, table):
    for flag, dest, kind, default in table:
        if kind == "flag":
            parser.add_option(flag, action="store_true", dest=dest, default=default)
        else:
            parser.add_option(flag, dest=dest, type={"str": "string"}.get(kind, kind), default=default)
    return parser


def get_lm_option_parser():
    return _add(_add(OptionParser(), _LM), _BUILD)


def get_img_options_parser():
    parser = _add(get_lm_option_parser(), _IMG)
    parser.set_default("batch", 20000)   # src/option_parser.py:54
    parser.set_default("mask_prob", 0.5)  # :62
    return parser
