"""Host-side argument validation of the C ABI, driven WITHOUT a GPU: every call below must be refused with IMT_ERR_ARG (-1)
before anything is enqueued.  The same file is the payload of the sanitizer run: ``tools/run_asan.sh`` builds
the host half of csrc/ with AddressSanitizer + UBSan (``make -C imagetranslate_amd/csrc asan``) and runs these tests and
tests/test_cabi.py against that library, so the descriptor walks, workspace-size arithmetic and layer-table indexing of
csrc/model.hip / gemm.hip / decode.hip execute under the sanitizers."""
import ctypes

import pytest

from imagetranslate_amd import _lib as L

ERR = -1


@pytest.fixture(scope="module")
def lib():
    return L.load()


def _stack(n_layers=2, d=128, heads=4, ff=512, decoder=False, dtype=1):
    arr = (L.LayerDesc * max(1, n_layers))()
    off = 4096
    for i in range(n_layers):
        for blk in (arr[i].self_attn, arr[i].cross_attn):
            for f, _ in L.AttnBlock._fields_:
                setattr(blk, f, off)
                off += 3 * d * d
        if not decoder:
            arr[i].cross_attn.qkv_w = -1
        for f in ("ff1_w", "ff1_b", "ff2_w", "ff2_b", "ln2_g", "ln2_b", "cross_kv_w", "cross_kv_b"):
            setattr(arr[i], f, off)
            off += d * ff
    s = L.StackDesc()
    s.dtype, s.d, s.heads, s.ff, s.vocab, s.max_pos, s.n_types, s.n_layers = dtype, d, heads, ff, 1000, 64, 2, n_layers
    s.is_decoder, s.pad_id, s.ln_eps = int(decoder), 0, 1e-12
    s.emb_word, s.emb_pos, s.emb_type, s.emb_ln_g, s.emb_ln_b = 0, 1024, 2048, 3072, 3584
    s.layers = ctypes.cast(arr, ctypes.POINTER(L.LayerDesc))
    s.params = 0x1000  # never dereferenced on the host
    s._keep = arr
    return s


def _io(B=2, T=8, Tk=8):
    io = L.StackIO()
    io.B, io.T, io.Tk = B, T, Tk
    io.ids, io.out = 0x2000, 0x3000
    return io


def test_gemm_argument_checks(lib):
    a = L.GemmArgs()
    assert lib.imt_gemm(None, None) == ERR
    a.dtype = 9
    assert lib.imt_gemm(ctypes.byref(a), None) == ERR
    a.dtype, a.M, a.N, a.K = 1, -1, 8, 8
    assert lib.imt_gemm(ctypes.byref(a), None) == ERR and b"negative" in lib.imt_last_error()
    a.M = 8
    assert lib.imt_gemm(ctypes.byref(a), None) == ERR and b"null operand" in lib.imt_last_error()
    a.A, a.B, a.C = 0x1000, 0x2000, 0x3000
    a.lda, a.ldb, a.ldc = 7, 8, 8
    assert lib.imt_gemm(ctypes.byref(a), None) == ERR and b"lda" in lib.imt_last_error()
    a.lda, a.A = 8, 0x1004
    assert lib.imt_gemm(ctypes.byref(a), None) == ERR and b"aligned" in lib.imt_last_error()
    a.A, a.ldc = 0x1000, 6
    assert lib.imt_gemm(ctypes.byref(a), None) == ERR and b"ldc" in lib.imt_last_error()
    a.ldc, a.c_dtype = 8, 5
    assert lib.imt_gemm(ctypes.byref(a), None) == ERR
    a.c_dtype, a.split_k, a.bias = 1, 2, 0x4000   # split-K with an epilogue
    assert lib.imt_gemm(ctypes.byref(a), None) == ERR and b"split_k" in lib.imt_last_error()
    # an empty product is accepted without a launch
    z = L.GemmArgs()
    z.dtype, z.M, z.N, z.K = 1, 0, 8, 8
    assert lib.imt_gemm(ctypes.byref(z), None) == 0
    assert lib.imt_gemm_grouped_tn(None, 3, None) == ERR
    assert lib.imt_gemm_grouped_tn(ctypes.byref(z), 0, None) == 0


def test_stack_descriptor_checks(lib):
    s, io = _stack(), _io()
    assert lib.imt_stack_forward(None, ctypes.byref(io), None, 0, None) == ERR
    assert lib.imt_stack_forward(ctypes.byref(s), None, None, 0, None) == ERR
    for field, bad, msg in (("dtype", 7, b"dtype"), ("n_layers", 10 ** 6, b"layer table"), ("heads", 5, b"heads"), ("d", 96, b"head_dim"),
                            ("ff", 516, b"multiples of 8")):
        s = _stack()
        setattr(s, field, bad)
        assert lib.imt_stack_forward(ctypes.byref(s), ctypes.byref(io), None, 0, None) == ERR, field
        assert msg in lib.imt_last_error(), (field, lib.imt_last_error())
    s = _stack()
    for field, bad, msg in (("B", 0, b"empty"), ("T", 65, b"longer"), ("ids", 0, b"null tensor")):
        io = _io()
        setattr(io, field, bad)
        assert lib.imt_stack_forward(ctypes.byref(s), ctypes.byref(io), None, 0, None) == ERR, field
        assert msg in lib.imt_last_error(), (field, lib.imt_last_error())
    dec = _stack(decoder=True)
    assert lib.imt_stack_forward(ctypes.byref(dec), ctypes.byref(_io()), None, 0, None) == ERR and b"encoder states" in lib.imt_last_error()
    # workspace: the size query walks the whole layer table; too small / misaligned buffers are refused
    need = lib.imt_stack_workspace_bytes(ctypes.byref(s), 2, 8, 8)
    assert need > 0 and need == lib.imt_stack_workspace_bytes(ctypes.byref(s), 2, 8, 8)
    assert lib.imt_stack_workspace_bytes(ctypes.byref(s), 4, 8, 8) > need
    io = _io()
    assert lib.imt_stack_forward(ctypes.byref(s), ctypes.byref(io), 0x10000, need - 1, None) == ERR and b"too small" in lib.imt_last_error()
    assert lib.imt_stack_forward(ctypes.byref(s), ctypes.byref(io), 0x10010, need, None) == ERR and b"aligned" in lib.imt_last_error()
    # backward: grads / d_out and the layer range
    assert lib.imt_stack_backward(ctypes.byref(s), ctypes.byref(io), 0x10000, need, 0, 2, None) == ERR and b"grads" in lib.imt_last_error()
    s.grads, io.d_out = 0x5000, 0x6000
    assert lib.imt_stack_backward(ctypes.byref(s), ctypes.byref(io), 0x10000, need, 2, 1, None) == ERR and b"layer range" in lib.imt_last_error()
    assert lib.imt_stack_backward(ctypes.byref(s), ctypes.byref(io), 0x10000, need, 0, 3, None) == ERR


def test_decode_descriptor_checks(lib):
    enc = _stack()
    assert lib.imt_decode_workspace_bytes(None, 4) < 0
    assert lib.imt_decode_begin(ctypes.byref(enc), 0x1000, 2, 8, 0x2000, None) == ERR and b"not a decoder" in lib.imt_last_error()
    dec = _stack(decoder=True)
    assert lib.imt_decode_workspace_bytes(ctypes.byref(dec), 4) > 0
    assert lib.imt_decode_self_cache_bytes(ctypes.byref(dec), 4, 16) > 0 and lib.imt_decode_cross_bytes(ctypes.byref(dec), 2, 8) > 0
    nocross = _stack(decoder=True)
    nocross._keep[1].cross_attn.qkv_w = -1
    assert lib.imt_decode_begin(ctypes.byref(nocross), 0x1000, 2, 8, 0x2000, None) == ERR and b"crossattention" in lib.imt_last_error()
    assert lib.imt_decode_begin(ctypes.byref(dec), None, 2, 8, None, None) == ERR
    io = L.DecodeIO()
    assert lib.imt_decode_step(ctypes.byref(dec), None, None, 0, None) == ERR
    io.R, io.rep, io.r_max, io.pos, io.t_max, io.Tk = 6, 4, 8, 0, 16, 8
    assert lib.imt_decode_step(ctypes.byref(dec), ctypes.byref(io), None, 0, None) == ERR and b"row counts" in lib.imt_last_error()
    io.R, io.pos = 8, 16
    assert lib.imt_decode_step(ctypes.byref(dec), ctypes.byref(io), None, 0, None) == ERR and b"outside the cache" in lib.imt_last_error()
    io.pos = 3
    assert lib.imt_decode_step(ctypes.byref(dec), ctypes.byref(io), None, 0, None) == ERR and b"null tensor" in lib.imt_last_error()
    # the one-launch step (bf16, hidden size 512) adds its per-layer hand-off buffers and barrier words to the workspace; a stack it does
    # not cover has nothing to check
    assert lib.imt_decode_check(ctypes.byref(dec), 0, 0x1000, None) == ERR and lib.imt_decode_check(ctypes.byref(dec), 8, None, None) == ERR
    assert lib.imt_decode_check(ctypes.byref(enc), 8, 0x1000, None) == ERR and b"not a decoder" in lib.imt_last_error()
    assert lib.imt_decode_check(ctypes.byref(dec), 8, 0x1000, None) == 0
    small = lib.imt_decode_workspace_bytes(ctypes.byref(_stack(n_layers=2, d=512, heads=8, ff=2048, decoder=True, dtype=0)), 320)
    fused = lib.imt_decode_workspace_bytes(ctypes.byref(_stack(n_layers=2, d=512, heads=8, ff=2048, decoder=True, dtype=1)), 320)
    six = lib.imt_decode_workspace_bytes(ctypes.byref(_stack(n_layers=6, d=512, heads=8, ff=2048, decoder=True, dtype=1)), 320)
    per_layer = 320 * (6 * 512 * 2 + 2048 * 2 + 3 * 512 * 4)
    assert fused - small // 2 >= 2 * per_layer and six - fused == 4 * per_layer, (small, fused, six, per_layer)
    ref = lib.imt_decode_workspace_bytes(ctypes.byref(_stack(n_layers=3, d=768, heads=12, ff=3072, decoder=True, dtype=1)), 320)
    ref1 = lib.imt_decode_workspace_bytes(ctypes.byref(_stack(n_layers=2, d=768, heads=12, ff=3072, decoder=True, dtype=1)), 320)
    assert ref - ref1 == 320 * (6 * 768 * 2 + 3072 * 2 + 3 * 768 * 4)   # the reference's default shape takes the one-launch step too
    odd = lib.imt_decode_workspace_bytes(ctypes.byref(_stack(n_layers=3, d=1024, heads=16, ff=4096, decoder=True, dtype=1)), 320)
    odd1 = lib.imt_decode_workspace_bytes(ctypes.byref(_stack(n_layers=2, d=1024, heads=16, ff=4096, decoder=True, dtype=1)), 320)
    assert odd == odd1                                                   # other shapes: the chain, no per-layer buffers


def test_row_kernel_argument_checks(lib):
    assert lib.imt_layernorm_fwd(1, None, None, None, None, None, None, 4, 6, 1e-12, 0.0, 0, None) == ERR
    assert lib.imt_layernorm_fwd(9, 0x1000, 0x1000, 0x1000, 0x1000, None, None, 4, 8, 1e-12, 0.0, 0, None) == ERR
    assert lib.imt_abi_sizeof(b"imt_gemm_args") == ctypes.sizeof(L.GemmArgs)
    assert lib.imt_abi_sizeof(None) == -1
