"""Tensor-level wrappers over the C ABI (no autograd here): each function takes torch CUDA tensors, passes
raw device pointers / sizes / the current HIP stream to ``libimt_hip.so`` and returns torch tensors that own
the outputs.  PyTorch supplies memory and streams only.
"""
import ctypes
import os
import math

import torch

from . import _lib as L
from ._lib import IMT_AUX_DGELU, IMT_AUX_GELU_FWD, IMT_AUX_NONE, IMT_AUX_SPLITK_WS, IMT_BF16, IMT_F32, IMT_NN, IMT_NT, IMT_TN  # noqa: F401


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return IMT_F32
    if t.dtype == torch.bfloat16:
        return IMT_BF16
    raise TypeError("imagetranslate_amd supports float32 and bfloat16 tensors, got %s" % t.dtype)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.ImtError("imagetranslate_amd HIP ops need tensors on the GPU (no CPU fallback)")


def alloc_rows(n_rows, width, dtype, device):
    """[n_rows, width] view of a zero-initialised buffer whose leading dimension is rounded up to 8 elements: rows stay
    16-byte aligned for the vector loads/stores when ``width`` (e.g. a tokenizer's vocabulary size) is not a multiple
    of 8, and the pad columns hold zeros."""
    padded = (width + 7) // 8 * 8
    if padded == width:
        return torch.empty((n_rows, width), device=device, dtype=dtype)
    return torch.zeros((n_rows, padded), device=device, dtype=dtype)[:, :width]


def rows16(t):
    """A 2-D tensor as rows the kernels can address with 16-byte vector accesses: unit inner stride and a leading
    dimension that is a multiple of 8 elements (already true for the padded buffers of alloc_rows); copies otherwise."""
    if t.dim() == 2 and t.stride(1) == 1 and (t.stride(0) % 8 == 0 or t.shape[0] <= 1) and t.data_ptr() % 16 == 0:
        return t
    out = alloc_rows(t.shape[0], t.shape[1], t.dtype, t.device)
    out.copy_(t)
    return out


def _rowmajor(t):
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D view expected"
    return t.stride(0)


def gemm(A, B, layout, *, out=None, out_dtype=None, bias=None, resid=None, aux=None, aux_mode=IMT_AUX_NONE,
         accumulate=False, split_k=1, alpha=1.0, dropout_p=0.0, dropout_seed=0, alpha_dev=None, a_colsum=None,
         force_general=False, force_pipeline=False, ln=None, splitk_ws=None, _launch=True):
    """C = epilogue(op(A) op(B)); see include/imt_hip.h:imt_gemm.  ``splitk_ws``: fp32 scratch (``splitk_workspace(device)``)
    that lets a product with few output tiles and a long K run as K ranges + one epilogue launch."""
    _req_cuda(A, B, out, bias, resid, aux)
    if layout == IMT_NT:
        M, K = A.shape; N = B.shape[0]; assert B.shape[1] == K
    elif layout == IMT_NN:
        M, K = A.shape; N = B.shape[1]; assert B.shape[0] == K
    else:
        K, M = A.shape; N = B.shape[1]; assert B.shape[0] == K
    if out is None:
        out = alloc_rows(M, N, out_dtype or A.dtype, A.device)
        assert not accumulate and split_k == 1
    a = L.GemmArgs()
    a.dtype, a.layout = dt(A), layout
    a.M, a.N, a.K = M, N, K
    a.A, a.lda = A.data_ptr(), _rowmajor(A)
    a.B, a.ldb = B.data_ptr(), _rowmajor(B)
    a.C, a.ldc = out.data_ptr(), _rowmajor(out)
    a.c_dtype = dt(out)
    a.accumulate = int(accumulate)
    a.bias = bias.data_ptr() if bias is not None else None
    a.resid, a.ldr = (resid.data_ptr(), _rowmajor(resid)) if resid is not None else (None, 0)
    a.aux, a.ldaux = (aux.data_ptr(), _rowmajor(aux)) if aux is not None else (None, 0)
    a.aux_mode, a.split_k = aux_mode, split_k
    a.alpha, a.dropout_p, a.dropout_seed = alpha, dropout_p, dropout_seed
    a.alpha_dev = alpha_dev.data_ptr() if alpha_dev is not None else None
    a.a_colsum = a_colsum.data_ptr() if a_colsum is not None else None
    a.force_general = int(force_general)
    a.force_pipeline = int(force_pipeline)
    if splitk_ws is not None:
        _req_cuda(splitk_ws)
        a.splitk_ws, a.splitk_ws_bytes = splitk_ws.data_ptr(), splitk_ws.numel() * splitk_ws.element_size()
    if ln is not None:
        # ln = dict(gamma, beta, out, mean, rstd, tickets (int32, zero), eps): LayerNorm of the rows of `out` in the same call
        _req_cuda(ln["gamma"], ln["beta"], ln["out"], ln["mean"], ln["rstd"], ln["tickets"])
        assert ln["tickets"].dtype == torch.int32 and ln["tickets"].numel() >= (M + 127) // 128
        a.ln_gamma, a.ln_beta = ln["gamma"].data_ptr(), ln["beta"].data_ptr()
        a.ln_out, a.ld_ln = ln["out"].data_ptr(), _rowmajor(ln["out"])
        a.ln_mean, a.ln_rstd = ln["mean"].data_ptr(), ln["rstd"].data_ptr()
        a.ln_tickets, a.ln_eps = ln["tickets"].data_ptr(), float(ln.get("eps", 1e-12))
    if not _launch:
        return a
    L.check(L.load().imt_gemm(ctypes.byref(a), _stream()), "imt_gemm")
    return out


_SPLITK_WS = {}


def splitk_workspace(device):
    """One fp32 scratch per device for imt_gemm's split-K slab mode (stream-ordered use: one product at a time)."""
    key = (device.type, device.index)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = torch.empty(int(L.load().imt_gemm_splitk_ws_bytes()) // 4, device=device, dtype=torch.float32)
        _SPLITK_WS[key] = ws
    return ws


def gemm_grouped_tn(problems):
    """problems: list of dicts(A=dy[K,M], B=x[K,N], out=fp32 grad [M,N], a_colsum=optional) -> one launch."""
    arr = (L.GemmArgs * len(problems))()
    for i, pr in enumerate(problems):
        arr[i] = gemm(pr["A"], pr["B"], IMT_TN, out=pr["out"], accumulate=True, a_colsum=pr.get("a_colsum"), _launch=False)
    L.check(L.load().imt_gemm_grouped_tn(arr, len(problems), _stream()), "imt_gemm_grouped_tn")


def gemm_bias_residual_ln(x, w, bias, resid, gamma, beta, eps=1e-12, dropout_p=0.0, dropout_seed=0, want_pre_ln=True):
    """out = LayerNorm(dropout(x w^T + bias) + resid) in one launch (imt_gemm_bias_residual_ln); returns
    (out, pre_ln or None, mean, rstd)."""
    _req_cuda(x, w, bias, resid, gamma, beta)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K
    out = torch.empty((M, N), device=x.device, dtype=x.dtype)
    pre = torch.empty((M, N), device=x.device, dtype=x.dtype) if want_pre_ln else None
    mean = torch.empty(M, device=x.device, dtype=torch.float32)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    L.check(L.load().imt_gemm_bias_residual_ln(dt(x), _p(x), _rowmajor(x), _p(w), _rowmajor(w), _p(bias), _p(resid),
                                               _rowmajor(resid) if resid is not None else 0, _p(gamma), _p(beta), _p(pre), _p(out), N,
                                               _p(mean), _p(rstd), M, N, K, eps, dropout_p, dropout_seed, _stream()),
            "imt_gemm_bias_residual_ln")
    return out, pre, mean, rstd


def colsum(X, out, scale_dev=None):
    _req_cuda(X, out)
    L.check(L.load().imt_colsum(dt(X), _p(X), _rowmajor(X), X.shape[0], X.shape[1], _p(out), _p(scale_dev), _stream()),
            "imt_colsum")
    return out


def layernorm_fwd(x, gamma, beta, eps=1e-12, dropout_p=0.0, dropout_seed=0):
    _req_cuda(x, gamma, beta)
    rows, d = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    L.check(L.load().imt_layernorm_fwd(dt(x), _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, d, eps,
                                       dropout_p, dropout_seed, _stream()), "imt_layernorm_fwd")
    return y, mean, rstd


def add_layernorm_fwd(x, resid, gamma, beta, eps=1e-12, dropout_p=0.0, dropout_seed=0):
    """(y, x + resid, mean, rstd) with y = dropout(LayerNorm(x + resid)) in one launch (imt_add_layernorm_fwd)."""
    _req_cuda(x, resid, gamma, beta)
    rows, d = x.shape
    y, ssum = torch.empty_like(x), torch.empty_like(x)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    L.check(L.load().imt_add_layernorm_fwd(dt(x), _p(x), _p(resid), _p(gamma), _p(beta), _p(ssum), _p(y), _p(mean), _p(rstd), rows, d,
                                           eps, dropout_p, dropout_seed, _stream()), "imt_add_layernorm_fwd")
    return y, ssum, mean, rstd


def embed_ln_fwd(ids, pos_ids, type_ids, word, pos, typ, gamma, beta, seq_len, eps=1e-12, dropout_p=0.0, dropout_seed=0):
    """BertEmbeddings in one launch (imt_embed_ln_fwd): (y, sum, mean, rstd)."""
    _req_cuda(ids, word, gamma, beta)
    n, d = ids.numel(), word.shape[1]
    y = torch.empty((n, d), device=word.device, dtype=word.dtype)
    ssum = torch.empty_like(y)
    mean = torch.empty(n, device=word.device, dtype=torch.float32)
    rstd = torch.empty(n, device=word.device, dtype=torch.float32)
    L.check(L.load().imt_embed_ln_fwd(dt(word), _p(ids), _p(pos_ids), _p(type_ids), _p(word), _p(pos), _p(typ), _p(gamma), _p(beta),
                                      _p(ssum), _p(y), _p(mean), _p(rstd), n, seq_len, d, word.shape[0], pos.shape[0], typ.shape[0], eps,
                                      dropout_p, dropout_seed, _stream()), "imt_embed_ln_fwd")
    return y, ssum, mean, rstd


LN_PARTIAL_COPIES = 32  # IMT_LN_PARTIAL_COPIES (include/imt_hip.h)


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta, y_dropout_p=0.0, y_dropout_seed=0, want_dx_drop=False,
                  dx_dropout_p=0.0, dx_dropout_seed=0, partial_ws=None):
    """partial_ws: optional ZEROED fp32 [LN_PARTIAL_COPIES, 2, d] buffer -- the column sums go there instead of into
    dgamma / dbeta (fold with ln_partial_reduce); None = straight atomics into dgamma / dbeta."""
    _req_cuda(dy, x, gamma, dgamma, dbeta, partial_ws)
    rows, d = x.shape
    dx = torch.empty_like(x)
    dx_drop = torch.empty_like(x) if want_dx_drop else None
    ws = partial_ws
    if ws is not None:
        assert ws.dtype == torch.float32 and ws.is_contiguous() and ws.numel() == LN_PARTIAL_COPIES * 2 * d
    L.check(L.load().imt_layernorm_bwd(dt(x), _p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dgamma),
                                       _p(dbeta), rows, d, y_dropout_p, y_dropout_seed, _p(dx_drop), dx_dropout_p,
                                       dx_dropout_seed, _p(ws), _stream()), "imt_layernorm_bwd")
    return (dx, dx_drop) if want_dx_drop else dx


def ln_partial_reduce(partials, grads, dgamma_offsets, dbeta_offsets):
    """grads[dgamma_offsets[i] + c] += sum over the copies of partials[i, :, 0, c] (likewise dbeta with [i, :, 1, c]);
    partials: fp32 [n, LN_PARTIAL_COPIES, 2, d]; a negative dgamma offset skips that buffer.  One launch."""
    _req_cuda(partials, grads)
    n, copies, two, d = partials.shape
    assert copies == LN_PARTIAL_COPIES and two == 2 and partials.is_contiguous() and grads.dtype == torch.float32
    g = (ctypes.c_int64 * n)(*[int(v) for v in dgamma_offsets])
    b = (ctypes.c_int64 * n)(*[int(v) for v in dbeta_offsets])
    L.check(L.load().imt_ln_partial_reduce(_p(partials), n, d, g, b, _p(grads), _stream()), "imt_ln_partial_reduce")


def embed_fwd(ids, pos_ids, type_ids, word, pos, typ, seq_len):
    _req_cuda(ids, word)
    n = ids.numel()
    d = word.shape[1]
    out = torch.empty((n, d), device=word.device, dtype=word.dtype)
    L.check(L.load().imt_embed_fwd(dt(word), _p(ids), _p(pos_ids), _p(type_ids), _p(word), _p(pos), _p(typ), _p(out), n,
                                   seq_len, d, word.shape[0], pos.shape[0], typ.shape[0], _stream()), "imt_embed_fwd")
    return out


def embed_bwd(ids, pos_ids, type_ids, dsum, dword, dpos, dtype_tab, seq_len, pad_id):
    _req_cuda(ids, dsum, dword, dpos, dtype_tab)
    n, d = dsum.shape
    L.check(L.load().imt_embed_bwd(dt(dsum), _p(ids), _p(pos_ids), _p(type_ids), _p(dsum), _p(dword), _p(dpos),
                                   _p(dtype_tab), n, seq_len, d, pad_id, _stream()), "imt_embed_bwd")


def _attn_args(q, k, v, B, H, Tq, Tk, head_dim, key_mask, query_mask, mask3d, causal, scale, dropout_p, dropout_seed):
    a = L.AttnArgs()
    a.dtype = dt(q)
    a.B, a.H, a.Tq, a.Tk, a.head_dim = B, H, Tq, Tk, head_dim
    a.Q, a.ldq = q.data_ptr(), _rowmajor(q)
    a.K, a.ldk = k.data_ptr(), _rowmajor(k)
    a.V, a.ldv = v.data_ptr(), _rowmajor(v)
    a.key_mask = key_mask.data_ptr() if key_mask is not None else None
    a.query_mask = query_mask.data_ptr() if query_mask is not None else None
    a.mask3d = mask3d.data_ptr() if mask3d is not None else None
    a.causal = int(causal)
    a.scale = scale if scale is not None else 1.0 / math.sqrt(head_dim)
    a.dropout_p, a.dropout_seed = dropout_p, dropout_seed
    return a


def attention_fwd(q, k, v, B, H, Tq, Tk, head_dim, key_mask=None, query_mask=None, mask3d=None, causal=False,
                  scale=None, dropout_p=0.0, dropout_seed=0):
    """q: [B*Tq, >=H*head_dim] row-major view (heads merged), k/v: [B*Tk, ...]; masks uint8."""
    _req_cuda(q, k, v, key_mask, query_mask, mask3d)
    a = _attn_args(q, k, v, B, H, Tq, Tk, head_dim, key_mask, query_mask, mask3d, causal, scale, dropout_p, dropout_seed)
    o = torch.empty((B * Tq, H * head_dim), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, H, Tq), device=q.device, dtype=torch.float32)
    a.O, a.ldo, a.lse = o.data_ptr(), o.stride(0), lse.data_ptr()
    L.check(L.load().imt_attention_fwd(ctypes.byref(a), _stream()), "imt_attention_fwd")
    return o, lse


def attention_qkv_fwd(x, w_qkv, b_qkv, B, H, T, head_dim, key_mask=None, query_mask=None, causal=False, scale=None,
                      dropout_p=0.0, dropout_seed=0):
    """(qkv [B*T, 3d], o [B*T, d], lse) = q|k|v projection + self-attention forward in ONE launch (imt_attention_qkv_fwd);
    x [B*T, d], w_qkv [3d, d] (query rows first), b_qkv [3d] or None."""
    _req_cuda(x, w_qkv, b_qkv, key_mask, query_mask)
    d = H * head_dim
    qkv = torch.empty((B * T, 3 * d), device=x.device, dtype=x.dtype)
    a = _attn_args(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, T, T, head_dim, key_mask, query_mask, None, causal, scale,
                   dropout_p, dropout_seed)
    o = torch.empty((B * T, d), device=x.device, dtype=x.dtype)
    lse = torch.empty((B, H, T), device=x.device, dtype=torch.float32)
    a.O, a.ldo, a.lse = o.data_ptr(), _rowmajor(o), lse.data_ptr()
    L.check(L.load().imt_attention_qkv_fwd(ctypes.byref(a), _p(x), _rowmajor(x), _p(w_qkv), _p(b_qkv), d, _stream()),
            "imt_attention_qkv_fwd")
    return qkv, o, lse


def attention_bwd(do, q, k, v, o, lse, B, H, Tq, Tk, head_dim, key_mask=None, query_mask=None, mask3d=None,
                  causal=False, scale=None, dropout_p=0.0, dropout_seed=0, dq=None, dk=None, dv=None):
    _req_cuda(do, q, k, v, o, lse)
    a = _attn_args(q, k, v, B, H, Tq, Tk, head_dim, key_mask, query_mask, mask3d, causal, scale, dropout_p, dropout_seed)
    if dq is None:
        dq = torch.empty((B * Tq, H * head_dim), device=q.device, dtype=q.dtype)
    if dk is None:
        dk = torch.empty((B * Tk, H * head_dim), device=q.device, dtype=q.dtype)
    if dv is None:
        dv = torch.empty((B * Tk, H * head_dim), device=q.device, dtype=q.dtype)
    delta = torch.empty((B, H, Tq), device=q.device, dtype=torch.float32)
    a.O, a.ldo, a.lse = o.data_ptr(), _rowmajor(o), lse.data_ptr()
    a.dO, a.lddo = do.data_ptr(), _rowmajor(do)
    a.dQ, a.lddq = dq.data_ptr(), _rowmajor(dq)
    a.dK, a.lddk = dk.data_ptr(), _rowmajor(dk)
    a.dV, a.lddv = dv.data_ptr(), _rowmajor(dv)
    a.delta = delta.data_ptr()
    L.check(L.load().imt_attention_bwd(ctypes.byref(a), _stream()), "imt_attention_bwd")
    return dq, dk, dv


def gather_rows(x, idx):
    _req_cuda(x, idx)
    out = torch.empty((idx.numel(), x.shape[1]), device=x.device, dtype=x.dtype)
    L.check(L.load().imt_gather_rows(dt(x), _p(x), _rowmajor(x), _p(idx), _p(out), out.stride(0) if out.numel() else x.shape[1],
                                     idx.numel(), x.shape[1], _stream()), "imt_gather_rows")
    return out


def scatter_rows(dout, idx, dx):
    _req_cuda(dout, idx, dx)
    L.check(L.load().imt_scatter_rows(dt(dout), _p(dout), _rowmajor(dout) if dout.numel() else dx.shape[1], _p(idx), _p(dx),
                                      _rowmajor(dx), idx.numel(), dx.shape[1], _stream()), "imt_scatter_rows")
    return dx


def log_softmax_fwd(logits):
    _req_cuda(logits)
    N, V = logits.shape
    lp = alloc_rows(N, V, torch.float32, logits.device)
    lse = torch.empty(N, device=logits.device, dtype=torch.float32)
    L.check(L.load().imt_log_softmax_fwd(dt(logits), _p(logits), _rowmajor(logits) if N else V, _p(lp), lp.stride(0) if N else V, _p(lse), N, V,
                                         _stream()), "imt_log_softmax_fwd")
    return lp, lse


def log_softmax_bwd(dlp, lp, out_dtype):
    _req_cuda(dlp, lp)
    N, V = lp.shape
    out = alloc_rows(N, V, out_dtype, lp.device)
    lp, dlp = rows16(lp), rows16(dlp)
    L.check(L.load().imt_log_softmax_bwd(_p(dlp), dlp.stride(0) if N else V, _p(lp), lp.stride(0) if N else V, dt(out), _p(out),
                                         out.stride(0) if N else V, N, V, _stream()),
            "imt_log_softmax_bwd")
    return out


def smoothed_nll_fwd(lp, target, epsilon, ignore_index):
    _req_cuda(lp, target)
    N, V = lp.shape
    loss = torch.empty((N, 1), device=lp.device, dtype=torch.float32)
    L.check(L.load().imt_smoothed_nll_fwd(_p(lp), _rowmajor(lp) if N else V, _p(target), _p(loss), N, V, epsilon,
                                          ignore_index, _stream()), "imt_smoothed_nll_fwd")
    return loss


def smoothed_nll_bwd(dloss, target, V, epsilon, ignore_index):
    _req_cuda(dloss, target)
    N = target.numel()
    dlp = alloc_rows(N, V, torch.float32, dloss.device)
    L.check(L.load().imt_smoothed_nll_bwd(_p(dloss), _p(target), _p(dlp), dlp.stride(0) if N else V, N, V, epsilon, ignore_index, _stream()),
            "imt_smoothed_nll_bwd")
    return dlp


def xent_fused_fwd_bwd(logits, target, epsilon, ignore_index, grad_scale):
    """In place: logits <- dlogits; returns per-row loss."""
    _req_cuda(logits, target)
    N, V = logits.shape
    loss = torch.empty(N, device=logits.device, dtype=torch.float32)
    L.check(L.load().imt_xent_fused_fwd_bwd(dt(logits), _p(logits), _rowmajor(logits) if N else V, _p(target), _p(loss), N, V,
                                            epsilon, ignore_index, grad_scale, _stream()), "imt_xent_fused_fwd_bwd")
    return loss


def scaled_sum(x, scale: float):
    """scale * x.sum() as a 0-d fp32 tensor (one small deterministic kernel)."""
    _req_cuda(x)
    out = torch.empty((), device=x.device, dtype=torch.float32)
    L.check(L.load().imt_scaled_sum(_p(x), x.numel(), float(scale), _p(out), _stream()), "imt_scaled_sum")
    return out


def sumsq(g, out, ws=None):
    _req_cuda(g, out, ws)
    if ws is None:
        ws = torch.empty(1024, device=g.device, dtype=torch.float32)  # IMT_SUMSQ_WS_FLOATS
    L.check(L.load().imt_sumsq(_p(g), g.numel(), _p(out), _p(ws), _stream()), "imt_sumsq")
    return out


def clip_adam(p, g, m, v, p_bf16, sumsq_t, max_norm, grad_scale, lr, beta1, beta2, eps, step, zero_grad=True):
    _req_cuda(p, g, m, v)
    L.check(L.load().imt_clip_adam(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), _p(sumsq_t), max_norm, grad_scale,
                                   lr, beta1, beta2, eps, step, int(zero_grad), _stream()), "imt_clip_adam")


def clip_scale(g, sumsq_t, max_norm, grad_scale=1.0):
    """g *= grad_scale * min(1, max_norm / (grad_scale * sqrt(sumsq) + 1e-6)), in place (imt_clip_scale)."""
    _req_cuda(g, sumsq_t)
    L.check(L.load().imt_clip_scale(_p(g), g.numel(), _p(sumsq_t), max_norm, grad_scale, _stream()), "imt_clip_scale")
    return g


def cast_f32_to_bf16(src, dst):
    _req_cuda(src, dst)
    L.check(L.load().imt_cast_f32_to_bf16(_p(src), _p(dst), src.numel(), _stream()), "imt_cast_f32_to_bf16")
    return dst


def gated_mix(a, b, gate):
    _req_cuda(a, b, gate)
    out = torch.empty_like(a)
    L.check(L.load().imt_gated_mix(dt(a), _p(a), _p(b), _p(gate), _p(out), a.shape[0], a.shape[1], _stream()), "imt_gated_mix")
    return out


def add_rows_dropout(x, add=None, out_dtype=None, dropout_p=0.0, dropout_seed=0):
    """out[r, :] = dropout(x[r, :] + add[r % add.shape[0], :]) for a 2-D x (imt_add_rows_dropout); add may be None."""
    _req_cuda(x, add)
    assert x.dim() == 2 and x.is_contiguous()
    out_dtype = out_dtype or x.dtype
    out = torch.empty(x.shape, device=x.device, dtype=out_dtype)
    if add is not None:
        assert add.dtype == out_dtype and add.is_contiguous() and add.shape[1] == x.shape[1]
    L.check(L.load().imt_add_rows_dropout(dt(x), _p(x), dt(out), _p(out), _p(add), x.shape[0], x.shape[1],
                                          add.shape[0] if add is not None else 1, float(dropout_p), int(dropout_seed), _stream()),
            "imt_add_rows_dropout")
    return out


# ------------------------------------------------------------------------------------------- incremental decoding
def attention_decode(q, k_base, v_base, n_keys, heads, *, ld_row, ld_pos, rep=1, slots=None, key_mask=None, ldq=None,
                     rows=None, scale=None):
    """Single-query attention against a (slot-addressed) K/V cache; see include/imt_hip.h:imt_attention_decode.
    ``k_base`` / ``v_base`` are tensors whose data_ptr is the (row 0, position 0, head 0) element."""
    _req_cuda(q, k_base, v_base, slots, key_mask)
    R = rows if rows is not None else q.shape[0]
    d = q.shape[-1]
    a = L.AttnDecodeArgs()
    a.dtype, a.R, a.H, a.head_dim, a.n_keys, a.rep = dt(q), R, heads, d // heads, n_keys, rep
    a.Q, a.ldq = q.data_ptr(), (ldq if ldq is not None else q.stride(0))
    a.K, a.V, a.ld_row, a.ld_pos = k_base.data_ptr(), v_base.data_ptr(), ld_row, ld_pos
    if slots is not None:
        assert slots.dtype == torch.int32
        a.slots, a.ld_slots = slots.data_ptr(), slots.stride(0)
    if key_mask is not None:
        assert key_mask.dtype == torch.uint8
        a.key_mask, a.ld_mask = key_mask.data_ptr(), key_mask.stride(0)
    out = torch.empty((R, d), device=q.device, dtype=q.dtype)
    a.O, a.ldo = out.data_ptr(), d
    a.scale = scale if scale is not None else 1.0 / math.sqrt(d // heads)
    L.check(L.load().imt_attention_decode(ctypes.byref(a), _stream()), "imt_attention_decode")
    return out


def beam_step(args: "L.BeamArgs"):
    L.check(L.load().imt_beam_step(ctypes.byref(args), _stream()), "imt_beam_step")


def select_plan(mask, ids, col0: int = 1, count=None):
    """(idx int32 [n], targets int64 [n]) of the positions with mask[b, col0 + t] set; ONE small kernel + one 4-byte
    device->host read (the row count shapes everything downstream, so this is the step's only synchronisation).
    `count`: the number of set positions when the caller already knows it on the host (a data loader builds the masks on
    the CPU) -- no read-back, the step is then enqueued without any synchronisation.  It MUST equal
    mask[:, col0:].sum(); IMT_CHECK_COUNT=1 verifies it (with the read-back)."""
    _req_cuda(mask, ids)
    B, T = ids.shape
    T1 = T - col0
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8) if mask.is_contiguous() else mask.contiguous().view(torch.uint8)
    assert mask.dtype == torch.uint8 and mask.stride(1) == 1 and ids.stride(1) == 1 and ids.dtype == torch.int64
    n = B * max(T1, 0)
    idx = torch.empty(max(n, 1), dtype=torch.int32, device=ids.device)
    targets = torch.empty(max(n, 1), dtype=torch.int64, device=ids.device)
    count_dev = torch.empty(1, dtype=torch.int32, device=ids.device)
    if n == 0:
        return idx[:0], targets[:0]
    L.check(L.load().imt_select_plan(_p(mask), mask.stride(0), _p(ids), ids.stride(0), B, T1, col0, _p(idx), _p(targets), _p(count_dev),
                                     _stream()), "imt_select_plan")
    if count is None or os.environ.get("IMT_CHECK_COUNT"):
        k = int(count_dev.item())
        if count is not None and int(count) != k:
            raise L.ImtError("select_plan: count hint %d but the mask selects %d positions" % (int(count), k))
    else:
        k = int(count)
    return idx[:k], targets[:k]
