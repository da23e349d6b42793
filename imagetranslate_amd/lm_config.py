"""BERT hyper-parameters of the path -- mirror of the reference's ``src/lm_config.py:4-30``.

Deviation (documented): ``num_attention_heads`` is a knob (reference hard-codes 12, which makes
BASELINE's d=512/h=8 and d=128/h=4 configurations unconstructible); the default stays 12.
"""
from typing import Dict


# (field, value) pairs the reference fixes for every model (src/lm_config.py:4-21); the four model-size fields are
# filled in by get_config from the constructor arguments of Seq2Seq.
_FIXED = (("hidden_act", "gelu"), ("initializer_range", 0.02), ("max_position_embeddings", 512),
          ("hidden_dropout_prob", 0.1), ("attention_probs_dropout_prob", 0.1))


def get_config(vocab_size: int, pad_token_id: int, bos_token_id: int, eos_token_id: int, enc_layer: int = 6,
               embed_dim: int = 768, intermediate_dim: int = 3072, num_attention_heads: int = 12) -> Dict:
    """Plain dict in the shape transformers.BertConfig(**d) takes (what src/seq2seq.py:37 does with it)."""
    cfg = dict(_FIXED)
    cfg.update(hidden_size=embed_dim, intermediate_size=intermediate_dim, num_hidden_layers=enc_layer,
               num_attention_heads=num_attention_heads)
    cfg.update(vocab_size=vocab_size, pad_token_id=pad_token_id, bos_token_id=bos_token_id, eos_token_id=eos_token_id)
    return cfg


class BertConfig:
    """Attribute bag with the fields of transformers.BertConfig that the path reads (src/seq2seq.py:37)."""

    def __init__(self, **kw):
        self.layer_norm_eps = 1e-12
        self.type_vocab_size = 2
        self.is_decoder = False
        for k, v in kw.items():
            setattr(self, k, v)
        if self.hidden_size % self.num_attention_heads != 0:
            # same failure mode as HF BertSelfAttention.__init__
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (self.hidden_size, self.num_attention_heads))

    def to_dict(self):
        return dict(self.__dict__)

    def __eq__(self, other):
        return isinstance(other, BertConfig) and self.__dict__ == other.__dict__
