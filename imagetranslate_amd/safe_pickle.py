"""Reading the reference's two pickled side files -- ``mt_config`` (a 9-tuple of bool/int, src/seq2seq.py:183-196) and
the tokenizer directory's ``langs`` (Dict[str, int], src/textprocessor.py:42-45) -- without executing anything from the
file: an Unpickler whose ``find_class`` always refuses.  Builtin containers and scalars need no class lookup, so the
reference's files load unchanged; a pickle that names any global (the vehicle of pickle code execution) is rejected.
"""
import io
import pickle


class _NoGlobals(pickle.Unpickler):
    def find_class(self, module, name):
        raise pickle.UnpicklingError("refusing to resolve %s.%s: only plain containers and scalars are accepted" % (module, name))


def load_plain(fp):
    data = fp.read() if hasattr(fp, "read") else fp
    return _NoGlobals(io.BytesIO(data)).load()


def load_mt_config(path):
    """(lang_dec, use_proposals, enc_layer, dec_layer, embed_dim, intermediate_dim, tie_embed, resnet_depth, freeze_image)"""
    with open(path, "rb") as fp:
        cfg = load_plain(fp)
    if not isinstance(cfg, tuple) or len(cfg) != 9 or not all(isinstance(v, (bool, int)) for v in cfg):
        raise ValueError("%s: expected the reference's 9-tuple of bool/int, got %r" % (path, type(cfg)))
    return cfg


def load_langs(path):
    with open(path, "rb") as fp:
        langs = load_plain(fp)
    if not isinstance(langs, dict) or not all(isinstance(k, str) and isinstance(v, int) and not isinstance(v, bool)
                                              for k, v in langs.items()):
        raise ValueError("%s: expected Dict[str, int] of language tags" % path)
    return langs
