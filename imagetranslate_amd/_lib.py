"""ctypes binding of the C-ABI library ``libimt_hip.so`` (declared in ``include/imt_hip.h``).

The product path has NO fallback: if the shared library is missing or a symbol cannot be bound this module
raises at import-of-use time.  PyTorch is used by callers only for device memory and streams; no torch type
crosses the boundary (plain pointers and sizes only).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IMT_LIB") or os.path.join(_HERE, "libimt_hip.so")  # IMT_LIB: another build of the SAME library (A/B runs)

IMT_F32, IMT_BF16 = 0, 1
IMT_NT, IMT_NN, IMT_TN = 0, 1, 2
IMT_AUX_NONE, IMT_AUX_GELU_FWD, IMT_AUX_DGELU, IMT_AUX_SPLITK_WS = 0, 1, 2, 3


class GemmArgs(Structure):
    _fields_ = [
        ("dtype", c_int32), ("layout", c_int32),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("A", c_void_p), ("lda", c_int64),
        ("B", c_void_p), ("ldb", c_int64),
        ("C", c_void_p), ("ldc", c_int64),
        ("c_dtype", c_int32), ("accumulate", c_int32),
        ("bias", c_void_p),
        ("resid", c_void_p), ("ldr", c_int64),
        ("aux", c_void_p), ("ldaux", c_int64),
        ("aux_mode", c_int32), ("split_k", c_int32),
        ("alpha", c_float), ("dropout_p", c_float),
        ("dropout_seed", c_uint64),
        ("alpha_dev", c_void_p),
        ("a_colsum", c_void_p),
        ("force_general", c_int32), ("force_pipeline", c_int32),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p),
        ("ln_out", c_void_p), ("ld_ln", c_int64),
        ("ln_mean", c_void_p), ("ln_rstd", c_void_p),
        ("ln_tickets", c_void_p),
        ("ln_eps", c_float), ("reserved_ln", c_int32),
        ("splitk_ws", c_void_p), ("splitk_ws_bytes", c_int64),
    ]


class AttnArgs(Structure):
    _fields_ = [
        ("dtype", c_int32),
        ("B", c_int32), ("H", c_int32), ("Tq", c_int32), ("Tk", c_int32), ("head_dim", c_int32),
        ("Q", c_void_p), ("ldq", c_int64),
        ("K", c_void_p), ("ldk", c_int64),
        ("V", c_void_p), ("ldv", c_int64),
        ("O", c_void_p), ("ldo", c_int64),
        ("lse", c_void_p),
        ("key_mask", c_void_p), ("query_mask", c_void_p), ("mask3d", c_void_p),
        ("causal", c_int32), ("scale", c_float), ("dropout_p", c_float),
        ("dropout_seed", c_uint64),
        ("dO", c_void_p), ("lddo", c_int64),
        ("dQ", c_void_p), ("lddq", c_int64),
        ("dK", c_void_p), ("lddk", c_int64),
        ("dV", c_void_p), ("lddv", c_int64),
        ("delta", c_void_p),
    ]


class ProfRow(Structure):
    _fields_ = [("kind", ctypes.c_char * 48), ("launches", c_int64), ("total_ms", ctypes.c_double),
                ("flops", ctypes.c_double), ("bytes", ctypes.c_double)]


class AttnBlock(Structure):
    _fields_ = [("qkv_w", c_int64), ("qkv_b", c_int64), ("o_w", c_int64), ("o_b", c_int64), ("ln_g", c_int64),
                ("ln_b", c_int64)]


class LayerDesc(Structure):
    _fields_ = [("self_attn", AttnBlock), ("cross_attn", AttnBlock),
                ("ff1_w", c_int64), ("ff1_b", c_int64), ("ff2_w", c_int64), ("ff2_b", c_int64),
                ("ln2_g", c_int64), ("ln2_b", c_int64), ("cross_kv_w", c_int64), ("cross_kv_b", c_int64)]


class StackDesc(Structure):
    _fields_ = [
        ("dtype", c_int32),
        ("d", c_int32), ("heads", c_int32), ("ff", c_int32), ("vocab", c_int32), ("max_pos", c_int32),
        ("n_types", c_int32), ("n_layers", c_int32),
        ("is_decoder", c_int32), ("reserved", c_int32),
        ("pad_id", c_int64),
        ("ln_eps", c_float), ("hidden_dropout", c_float), ("attn_dropout", c_float), ("reserved_f", c_float),
        ("emb_word", c_int64), ("emb_pos", c_int64), ("emb_type", c_int64), ("emb_ln_g", c_int64), ("emb_ln_b", c_int64),
        ("layers", POINTER(LayerDesc)),
        ("params", c_void_p),
        ("grads", c_void_p),
    ]


class StackIO(Structure):
    _fields_ = [
        ("B", c_int32), ("T", c_int32), ("Tk", c_int32), ("training", c_int32),
        ("ids", c_void_p), ("type_ids", c_void_p), ("pos_ids", c_void_p),
        ("key_mask", c_void_p), ("query_mask", c_void_p), ("mask3d", c_void_p),
        ("causal", c_int32), ("reserved", c_int32),
        ("enc_states", c_void_p), ("enc_mask", c_void_p),
        ("out", c_void_p),
        ("dropout_seed", c_uint64),
        ("d_out", c_void_p), ("d_enc_states", c_void_p),
        ("wait_events", POINTER(c_void_p)), ("n_wait_events", c_int32), ("reserved2", c_int32),
    ]


class AttnDecodeArgs(Structure):
    _fields_ = [
        ("dtype", c_int32),
        ("R", c_int32), ("H", c_int32), ("head_dim", c_int32), ("n_keys", c_int32), ("rep", c_int32),
        ("Q", c_void_p), ("ldq", c_int64),
        ("K", c_void_p), ("V", c_void_p), ("ld_row", c_int64), ("ld_pos", c_int64),
        ("slots", c_void_p), ("ld_slots", c_int64),
        ("key_mask", c_void_p), ("ld_mask", c_int64),
        ("O", c_void_p), ("ldo", c_int64),
        ("scale", c_float), ("reserved", c_int32),
    ]


class DecodeIO(Structure):
    _fields_ = [
        ("R", c_int32), ("rep", c_int32), ("pos", c_int32), ("Tk", c_int32), ("t_max", c_int32), ("r_max", c_int32),
        ("ids", c_void_p), ("type_ids", c_void_p), ("pos_ids", c_void_p), ("slots", c_void_p), ("enc_mask", c_void_p),
        ("self_cache", c_void_p), ("cross_kv", c_void_p), ("out", c_void_p),
    ]


class BeamArgs(Structure):
    _fields_ = [
        ("B", c_int32), ("beam", c_int32), ("rep", c_int32), ("V", c_int32), ("step", c_int32), ("t_max", c_int32),
        ("logits", c_void_p), ("ld", c_int64),
        ("scores_in", c_void_p), ("sizes_in", c_void_p), ("eos_in", c_void_p), ("max_lens", c_void_p),
        ("hist_in", c_void_p), ("slots_in", c_void_p),
        ("len_penalty_ratio", c_float), ("reserved", c_int32),
        ("pad_idx", c_int64), ("eos", c_int64),
        ("cand_scores", c_void_p), ("cand_idx", c_void_p),
        ("scores_out", c_void_p), ("sizes_out", c_void_p), ("eos_out", c_void_p),
        ("hist_out", c_void_p), ("slots_out", c_void_p), ("parent_out", c_void_p), ("tokens_out", c_void_p),
        ("eos_count", c_void_p),
    ]


class MassArgs(Structure):
    _fields_ = [
        ("n_rows", c_int32), ("width", c_int32), ("recover_width", c_int32), ("n_special", c_int32), ("vocab", c_int32),
        ("mask_prob", c_float), ("seed", c_uint64), ("mask_id", c_int64), ("pad_id", c_int64),
        ("src_text", c_void_p), ("pad_indices", c_void_p), ("row_offsets", c_void_p), ("src_mask", c_void_p),
        ("to_recover", c_void_p), ("positions", c_void_p), ("targets", c_void_p),
    ]


# name -> (restype, argtypes); must list EVERY symbol include/imt_hip.h declares (tests/test_cabi.py checks)
_P = c_void_p
SIGNATURES = {
    "imt_version": (c_int, []),
    "imt_last_error": (c_char_p, []),
    "imt_gemm": (c_int, [POINTER(GemmArgs), _P]),
    "imt_gemm_grouped_tn": (c_int, [POINTER(GemmArgs), c_int, _P]),
    "imt_gemm_splitk_ws_bytes": (c_int64, []),
    "imt_gemm_bias_residual_ln_supported": (c_int, [c_int, c_int, c_int]),
    "imt_gemm_bias_residual_ln": (c_int, [c_int, _P, c_int64, _P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, c_int64, _P, _P, c_int, c_int,
                                          c_int, c_float, c_float, c_uint64, _P]),
    "imt_colsum": (c_int, [c_int, _P, c_int64, c_int, c_int, _P, _P, _P]),
    "imt_layernorm_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, c_uint64, _P]),
    "imt_add_layernorm_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, c_uint64, _P]),
    "imt_embed_ln_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                 c_float, c_uint64, _P]),
    "imt_layernorm_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_uint64, _P, c_float,
                                  c_uint64, _P, _P]),
    "imt_embed_fwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "imt_embed_bwd": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int64, _P]),
    "imt_attention_fwd": (c_int, [POINTER(AttnArgs), _P]),
    "imt_attention_bwd": (c_int, [POINTER(AttnArgs), _P]),
    "imt_attention_qkv_fwd_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "imt_attention_qkv_fwd": (c_int, [POINTER(AttnArgs), _P, c_int64, _P, _P, c_int, _P]),
    "imt_gather_rows": (c_int, [c_int, _P, c_int64, _P, _P, c_int64, c_int, c_int, _P]),
    "imt_scatter_rows": (c_int, [c_int, _P, c_int64, _P, _P, c_int64, c_int, c_int, _P]),
    "imt_log_softmax_fwd": (c_int, [c_int, _P, c_int64, _P, c_int64, _P, c_int, c_int, _P]),
    "imt_log_softmax_bwd": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, c_int64, c_int, c_int, _P]),
    "imt_smoothed_nll_fwd": (c_int, [_P, c_int64, _P, _P, c_int, c_int, c_float, c_int64, _P]),
    "imt_smoothed_nll_bwd": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_float, c_int64, _P]),
    "imt_xent_fused_fwd_bwd": (c_int, [c_int, _P, c_int64, _P, _P, c_int, c_int, c_float, c_int64, c_float, _P]),
    "imt_scaled_sum": (c_int, [_P, c_int, c_float, _P, _P]),
    "imt_sumsq": (c_int, [_P, c_int64, _P, _P, _P]),
    "imt_ln_partial_reduce": (c_int, [_P, c_int, c_int, _P, _P, _P, _P]),
    "imt_clip_adam": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, c_float, c_float, c_float, c_float, c_float, c_float,
                              c_int64, c_int, _P]),
    "imt_clip_scale": (c_int, [_P, c_int64, _P, c_float, c_float, _P]),
    "imt_cast_f32_to_bf16": (c_int, [_P, _P, c_int64, _P]),
    "imt_gated_mix": (c_int, [c_int, _P, _P, _P, _P, c_int64, c_int, _P]),
    "imt_add_rows_dropout": (c_int, [c_int, _P, c_int, _P, _P, c_int64, c_int, c_int, c_float, c_uint64, _P]),
    "imt_comm_unique_id_bytes": (c_int, []),
    "imt_comm_get_unique_id": (c_int, [_P]),
    "imt_comm_init": (c_int, [_P, c_int, c_int, POINTER(c_void_p)]),
    "imt_comm_allreduce": (c_int, [_P, _P, c_int64, c_int, _P]),
    "imt_comm_broadcast": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "imt_comm_destroy": (c_int, [_P]),
    "imt_set_gemm_share_cus": (c_int, [c_int]),
    "imt_debug_spin": (c_int, [c_int, c_int, c_int, c_int64, _P]),
    "imt_abi_sizeof": (c_int, [ctypes.c_char_p]),
    "imt_prof_enable": (c_int, [c_int]),
    "imt_prof_report": (c_int, [POINTER(ProfRow), c_int]),
    "imt_stack_workspace_bytes": (c_int64, [POINTER(StackDesc), c_int, c_int, c_int]),
    "imt_stack_forward": (c_int, [POINTER(StackDesc), POINTER(StackIO), _P, c_int64, _P]),
    "imt_stack_backward": (c_int, [POINTER(StackDesc), POINTER(StackIO), _P, c_int64, c_int, c_int, _P]),
    "imt_attention_decode": (c_int, [POINTER(AttnDecodeArgs), _P]),
    "imt_decode_workspace_bytes": (c_int64, [POINTER(StackDesc), c_int]),
    "imt_decode_self_cache_bytes": (c_int64, [POINTER(StackDesc), c_int, c_int]),
    "imt_decode_cross_bytes": (c_int64, [POINTER(StackDesc), c_int, c_int]),
    "imt_decode_begin": (c_int, [POINTER(StackDesc), _P, c_int, c_int, _P, _P]),
    "imt_decode_step": (c_int, [POINTER(StackDesc), POINTER(DecodeIO), _P, c_int64, _P]),
    "imt_decode_check": (c_int, [POINTER(StackDesc), c_int, _P, _P]),
    "imt_beam_step": (c_int, [POINTER(BeamArgs), _P]),
    "imt_select_plan": (c_int, [_P, c_int64, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, _P]),
    "imt_mass_mask": (c_int, [POINTER(MassArgs), _P]),
    "imt_mass_unmask": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
}

_lib = None


class ImtError(RuntimeError):
    pass


ABI_STRUCTS = {"imt_gemm_args": GemmArgs, "imt_attn_args": AttnArgs, "imt_prof_row": ProfRow, "imt_attn_block": AttnBlock,
               "imt_layer_desc": LayerDesc, "imt_stack_desc": StackDesc, "imt_stack_io": StackIO,
               "imt_attn_decode_args": AttnDecodeArgs, "imt_decode_io": DecodeIO, "imt_beam_args": BeamArgs,
               "imt_mass_args": MassArgs}


def load():
    """Load libimt_hip.so and bind every declared symbol.  Raises (never falls back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImtError(
            "imagetranslate_amd: HIP extension %s is missing -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # the structures of this binding must be the library's, field for field: compare sizes once (a library built from another
    # revision of include/imt_hip.h would otherwise receive shifted fields)
    for cname, cls in ABI_STRUCTS.items():
        want = lib.imt_abi_sizeof(cname.encode())
        if want != ctypes.sizeof(cls):
            raise ImtError("imagetranslate_amd: %s is %d bytes in %s but %d in this binding -- rebuild the library "
                           "(python -c 'import __graft_entry__ as g; g.build()')" % (cname, want, LIB_PATH, ctypes.sizeof(cls)))
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().imt_last_error()
        raise ImtError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))
