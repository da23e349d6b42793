"""GPU parity tests of every C-ABI kernel against plain fp32 torch (CPU) restatements of the same op.

fp32 mode: 1e-4 relative (north_star tolerance).  bf16 mode: inputs are rounded to bf16 first and the reference
is computed in fp32 from the rounded inputs; tolerance 2.5e-2 relative to the output's max magnitude.
All calls go through the C ABI (imagetranslate_amd.hip_ops -> ctypes -> libimt_hip.so).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import TOL, assert_close

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _mk(shape, dtype, dev, scale=1.0, gen=None):
    x = torch.randn(shape, generator=gen) * scale
    xq = x.to(dtype)
    return xq.to(dev), xq.float()  # device tensor, fp32 CPU copy of the (rounded) values


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_exact_integer_layout(cuda, dtype, layout):
    """Small-integer operands (exact in bf16 and in fp32 accumulation): result must be BIT-EXACT.
    Asymmetric data so that any swapped row/col or k mapping shows."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(1)
    for (M, N, K) in [(128, 128, 64), (144, 200, 72), (16, 8, 8), (300, 264, 136)]:
        A = torch.randint(-3, 4, (M, K), generator=g).float()
        B = torch.randint(-3, 4, (N, K), generator=g).float()
        A[:, 0] += torch.arange(M).float() % 5
        B[0, :] += torch.arange(K).float() % 3
        ref = A @ B.t()
        if layout == O.IMT_NT:
            a_in, b_in = A, B
        elif layout == O.IMT_NN:
            a_in, b_in = A, B.t().contiguous()
        else:
            a_in, b_in = A.t().contiguous(), B.t().contiguous()
        if (a_in.shape[1] % 8) or (b_in.shape[1] % 8):
            continue
        out = O.gemm(a_in.to(dtype).to(cuda), b_in.to(dtype).to(cuda), layout, out_dtype=torch.float32)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref), "layout %d dtype %s shape %s: max diff %g" % (
            layout, dtype, (M, N, K), float((out.cpu() - ref).abs().max()))


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_random(cuda, dtype, layout):
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(2)
    for (M, N, K) in [(256, 384, 512), (1000, 136, 264), (77, 1000, 128), (512, 512, 2048)]:
        A, Af = _mk((M, K), dtype, cuda, 1.0, g)
        B, Bf = _mk((N, K), dtype, cuda, 1.0, g)
        ref = Af @ Bf.t()
        if layout == O.IMT_NT:
            a_in, b_in = A, B
        elif layout == O.IMT_NN:
            a_in, b_in = A, B.t().contiguous()
        else:
            a_in, b_in = A.t().contiguous(), B.t().contiguous()
        if (a_in.shape[1] % 8) or (b_in.shape[1] % 8):
            continue
        out = O.gemm(a_in, b_in, layout)
        assert_close(out, ref, 1e-5 if dtype == torch.float32 else 1e-2, "gemm %s %s" % (layout, (M, N, K)))


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(cuda, dtype):
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(3)
    M, N, K = 200, 264, 136
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    A, Af = _mk((M, K), dtype, cuda, 1.0, g)
    B, Bf = _mk((N, K), dtype, cuda, 0.2, g)
    bias, biasf = _mk((N,), dtype, cuda, 1.0, g)
    R, Rf = _mk((M, N), dtype, cuda, 1.0, g)
    base = Af @ Bf.t()
    # bias
    assert_close(O.gemm(A, B, O.IMT_NT, bias=bias), base + biasf, tol, "bias")
    # bias + residual
    assert_close(O.gemm(A, B, O.IMT_NT, bias=bias, resid=R), base + biasf + Rf, tol, "bias+resid")
    # bias + gelu (aux = pre-activation)
    aux = torch.empty((M, N), device=cuda, dtype=dtype)
    h = O.gemm(A, B, O.IMT_NT, bias=bias, aux=aux, aux_mode=O.IMT_AUX_GELU_FWD)
    assert_close(aux, base + biasf, tol, "gelu aux")
    assert_close(h, F.gelu(base + biasf), tol, "gelu")
    # dgelu: out = acc * gelu'(aux)
    z = (base + biasf).to(dtype).float().requires_grad_(True)
    F.gelu(z).sum().backward()
    assert_close(O.gemm(A, B, O.IMT_NT, aux=aux, aux_mode=O.IMT_AUX_DGELU), base * z.grad, tol * 2, "dgelu")
    # alpha + accumulate into fp32 C
    C0 = torch.randn((M, N), generator=g)
    C = C0.clone().to(cuda)
    O.gemm(A, B, O.IMT_NT, out=C, accumulate=True, alpha=0.5)
    assert_close(C, C0 + 0.5 * base, tol, "alpha+accumulate f32")
    # split-K atomic accumulation (TN, the weight-gradient form): dW[N,K] += dY[M,N]^T X[M,K]
    G0 = torch.randn((N, K), generator=g)
    G = G0.clone().to(cuda)
    dY, dYf = _mk((M, N), dtype, cuda, 1.0, g)
    O.gemm(dY, A, O.IMT_TN, out=G, split_k=4)
    assert_close(G, G0 + dYf.t() @ Af, tol, "split-k TN")
    # strided views (sub-matrix of a wider buffer), as used for the fused QKV buffer
    wide, widef = _mk((M, 3 * K), dtype, cuda, 1.0, g)
    out = O.gemm(wide[:, K:2 * K], B, O.IMT_NT)
    assert_close(out, widef[:, K:2 * K] @ Bf.t(), tol, "strided A")
    # dropout epilogue: kept elements scaled by 1/(1-p), mask deterministic in (seed, index)
    p = 0.25
    d1 = O.gemm(A, B, O.IMT_NT, bias=bias, dropout_p=p, dropout_seed=1234).float().cpu()
    d2 = O.gemm(A, B, O.IMT_NT, bias=bias, dropout_p=p, dropout_seed=1234).float().cpu()
    assert torch.equal(d1, d2)
    keep = d1 != 0
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 0.02, frac
    full = (base + biasf) / (1 - p)
    assert_close(d1[keep], full[keep], tol, "dropout kept values")


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_pipelined_vs_general_kernel(cuda, dtype, layout):
    """The LDS-DMA pipelined main loop (whole K tiles) and the register-staged general kernel must agree BIT-EXACTLY
    on integer data, for 2..many K tiles, ragged M/N, strided operand views and split-K."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(11)
    bk = 64 if dtype == torch.bfloat16 else 32
    for (M, N, nk, sk) in [(128, 128, 2, 1), (200, 136, 3, 1), (256, 264, 9, 1), (136, 128, 16, 4), (1000, 72, 5, 1)]:
        K = nk * bk
        A = torch.randint(-3, 4, (M, K), generator=g).float()
        B = torch.randint(-3, 4, (N, K), generator=g).float()
        A[:, 1] += torch.arange(M).float() % 7
        ref = A @ B.t()
        if layout == O.IMT_NT:
            a_in, b_in = A, B
        elif layout == O.IMT_NN:
            a_in, b_in = A, B.t().contiguous()
        else:
            a_in, b_in = A.t().contiguous(), B.t().contiguous()
        if (a_in.shape[1] % 8) or (b_in.shape[1] % 8):
            continue
        # embed the operands in wider buffers (strided views, like q/k/v inside the fused qkv buffer)
        wa = torch.full((a_in.shape[0], a_in.shape[1] + 16), 99.0); wa[:, 8:8 + a_in.shape[1]] = a_in
        wb = torch.full((b_in.shape[0], b_in.shape[1] + 24), -77.0); wb[:, 16:16 + b_in.shape[1]] = b_in
        da = wa.to(dtype).to(cuda)[:, 8:8 + a_in.shape[1]]
        db = wb.to(dtype).to(cuda)[:, 16:16 + b_in.shape[1]]
        outs = []
        for force in (2, 1, 3, 5, 6):  # LDS-DMA ring, register double buffer, single buffer, persistent wave-specialised, 256-tile
            out = torch.zeros((M, N), device=cuda, dtype=torch.float32)
            O.gemm(da, db, layout, out=out, split_k=sk if layout == O.IMT_TN else 1,
                   accumulate=False, force_general=force)
            outs.append(out.cpu())
        assert torch.equal(outs[0], ref), "pipelined kernel wrong: layout %d %s max diff %g" % (
            layout, (M, N, K), float((outs[0] - ref).abs().max()))
        assert torch.equal(outs[1], ref), "general kernel wrong"
        assert torch.equal(outs[2], ref), "single-buffer kernel wrong"
        assert torch.equal(outs[3], ref), "wave-specialised kernel wrong"
        assert torch.equal(outs[4], ref), "256 x 256-tile kernel wrong"


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues_agree_across_kernel_variants(cuda, dtype, layout):
    """Every kernel variant (single buffer, register double buffer, persistent wave-specialised with several tiles
    per workgroup, 256 x 256 tiles with their direct epilogue on full tiles) must produce BIT-IDENTICAL results for every epilogue the train step uses, on ragged M / N."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(21)
    bk = 64 if dtype == torch.bfloat16 else 32
    for (M, N, nk) in [(128 * 19 + 40, 128 * 15 + 72, 4), (300, 520, 9), (128 * 34, 256, 17)]:
        K = nk * bk
        A = (torch.randn(M, K, generator=g) * 0.5).to(dtype).to(cuda)
        B = (torch.randn((N, K) if layout == O.IMT_NT else (K, N), generator=g) * 0.5).to(dtype).to(cuda)
        bias = torch.randn(N, generator=g).to(dtype).to(cuda)
        resid = torch.randn(M, N, generator=g).to(dtype).to(cuda)
        z = torch.randn(M, N, generator=g).to(dtype).to(cuda)
        cases = [dict(), dict(bias=bias), dict(bias=bias, aux_mode=O.IMT_AUX_GELU_FWD), dict(aux_mode=O.IMT_AUX_DGELU),
                 dict(bias=bias, dropout_p=0.1, dropout_seed=77, resid=resid), dict(resid=resid), dict(alpha=0.25, resid=resid)]
        for kw in cases:
            outs, auxs = [], []
            for force in (3, 1, 5, 6):
                aux = None
                if kw.get("aux_mode") == O.IMT_AUX_GELU_FWD:
                    aux = torch.zeros(M, N, device=cuda, dtype=dtype)
                elif kw.get("aux_mode") == O.IMT_AUX_DGELU:
                    aux = z
                out = torch.zeros(M, N, device=cuda, dtype=dtype)
                O.gemm(A, B, layout, out=out, aux=aux, force_general=force, **kw)
                outs.append(out.float().cpu())
                auxs.append(None if aux is None else aux.float().cpu())
            for o in outs[1:]:
                assert torch.equal(outs[0], o), "epilogue %s differs (max %g)" % (sorted(kw), float((outs[0] - o).abs().max()))
            for x in auxs[1:]:
                assert auxs[0] is None or torch.equal(auxs[0], x)
        # fp32 output with accumulate (the form the runtime uses for C += ...)
        c0 = torch.randn(M, N, generator=g).to(cuda)
        o3, o5, o6 = c0.clone(), c0.clone(), c0.clone()
        O.gemm(A, B, layout, out=o3, accumulate=True, force_general=3)
        O.gemm(A, B, layout, out=o5, accumulate=True, force_general=5)
        O.gemm(A, B, layout, out=o6, accumulate=True, force_general=6)
        assert torch.equal(o3.cpu(), o5.cpu()) and torch.equal(o3.cpu(), o6.cpu())


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_long_ragged_k_split(cuda, dtype, layout):
    """K >= 16 tiles but not a whole number of tiles (the vocabulary dimension): imt_gemm runs the ragged tail and
    the whole-tile body as two launches; result == one product, including bias / residual / accumulate / alpha."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(31)
    bk = 64 if dtype == torch.bfloat16 else 32
    M, N, K = 264, 200, 17 * bk + 24
    A = torch.randint(-2, 3, (M, K), generator=g).float()
    B = torch.randint(-2, 3, (N, K), generator=g).float()
    ref = A @ B.t()
    a_in, b_in = (A, B) if layout == O.IMT_NT else ((A, B.t().contiguous()) if layout == O.IMT_NN else (A.t().contiguous(), B.t().contiguous()))
    da, db = a_in.to(dtype).to(cuda), b_in.to(dtype).to(cuda)
    out = torch.zeros(M, N, device=cuda, dtype=torch.float32)
    O.gemm(da, db, layout, out=out)
    assert torch.equal(out.cpu(), ref), "plain product (integer data, exact)"
    c0 = torch.randint(-5, 6, (M, N), generator=g).float()
    out = c0.clone().to(cuda)
    O.gemm(da, db, layout, out=out, accumulate=True, alpha=0.5)
    assert torch.equal(out.cpu(), c0 + 0.5 * ref), "alpha + accumulate"
    if layout != O.IMT_TN:
        bias = torch.randint(-3, 4, (N,), generator=g).float()
        resid = torch.randint(-3, 4, (M, N), generator=g).float()
        out = O.gemm(da, db, layout, bias=bias.to(dtype).to(cuda), resid=resid.to(dtype).to(cuda), out_dtype=torch.float32)
        assert torch.equal(out.cpu(), ref + bias + resid), "bias + residual applied exactly once"


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_tn_fused_bias_gradient(cuda, dtype):
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(12)
    for (tokens, out_f, in_f, sk, force) in [(512, 200, 136, 1, False), (512, 200, 136, 4, False), (300, 264, 72, 1, True),
                                             (1024, 1000, 128, 2, False), (512, 200, 136, 1, 5), (512, 200, 136, 1, 6),
                                             (1024, 1000, 392, 1, 6)]:  # 5 / 6: persistent and 256-tile kernels
        dy, dyf = _mk((tokens, out_f), dtype, cuda, 1.0, g)
        x, xf = _mk((tokens, in_f), dtype, cuda, 1.0, g)
        gw0 = torch.randn((out_f, in_f), generator=g); gb0 = torch.randn(out_f, generator=g)
        gw, gb = gw0.clone().to(cuda), gb0.clone().to(cuda)
        scale = torch.tensor([0.5], device=cuda)
        O.gemm(dy, x, O.IMT_TN, out=gw, split_k=sk, accumulate=(sk == 1), a_colsum=gb, alpha_dev=scale, force_general=force)
        tol = 2e-5 if dtype == torch.float32 else 1e-2
        assert_close(gw, gw0 + 0.5 * dyf.t() @ xf, tol, "dW")
        assert_close(gb, gb0 + 0.5 * dyf.sum(0), tol, "db (fused colsum)")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_grouped_weight_gradients(cuda, dtype):
    """imt_gemm_grouped_tn: the 4..7 weight-gradient GEMMs of a layer in one launch == the individual results; token counts
    that are not a whole number of K tiles (520, 992: real batches) take the same launch -- the rows of the ragged last K tile
    lie past the operands' valid bytes and the LDS-DMA range check zero-fills them (the memory behind the operands here is
    whatever the allocator left there)."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(13)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    torch.full((1 << 22,), 7.0, device=cuda)  # poison recycled allocator blocks: a read past an operand would not see zeros
    for tokens in (512, 520, 992):
        shapes = [(384, 128), (128, 128), (512, 128), (128, 512), (200, 136)]
        probs, refs = [], []
        for (out_f, in_f) in shapes:
            dy, dyf = _mk((tokens, out_f), dtype, cuda, 1.0, g)
            x, xf = _mk((tokens, in_f), dtype, cuda, 1.0, g)
            gw0 = torch.randn((out_f, in_f), generator=g); gb0 = torch.randn(out_f, generator=g)
            gw, gb = gw0.clone().to(cuda), gb0.clone().to(cuda)
            probs.append(dict(A=dy, B=x, out=gw, a_colsum=gb))
            refs.append((gw0 + dyf.t() @ xf, gb0 + dyf.sum(0)))
        O.gemm_grouped_tn(probs)
        for pr, (rw, rb) in zip(probs, refs):
            assert_close(pr["out"], rw, tol, "grouped dW (tokens=%d)" % tokens)
            assert_close(pr["a_colsum"], rb, tol, "grouped db (tokens=%d)" % tokens)
    # one launch also for the ragged counts
    from imagetranslate_amd import _lib as L
    lib = L.load()
    torch.cuda.synchronize()
    lib.imt_prof_enable(1)
    O.gemm_grouped_tn(probs)
    torch.cuda.synchronize()
    rows = (L.ProfRow * 16)()
    n = lib.imt_prof_report(rows, 16)
    lib.imt_prof_enable(0)
    assert n == 1 and rows[0].kind.decode().endswith("tn_grouped") and rows[0].launches == 1


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_tn_ragged_k_on_the_dma_kernels(cuda, dtype):
    """Weight-gradient products (TN: K = token count) with a K that is not a whole number of tiles on the persistent and the
    256-tile kernels (variants 5 / 6): the ragged last K tile is zero-filled by the descriptors' range check."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(31)
    torch.full((1 << 22,), 5.0, device=cuda)
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    for variant, (K, M, N) in ((5, (1000, 256, 384)), (6, (2100, 512, 256)), (5, (72, 128, 128))):
        dy, dyf = _mk((K, M), dtype, cuda, 1.0, g)
        x, xf = _mk((K, N), dtype, cuda, 1.0, g)
        gw0 = torch.randn((M, N), generator=g); gb0 = torch.randn(M, generator=g)
        gw, gb = gw0.clone().to(cuda), gb0.clone().to(cuda)
        O.gemm(dy, x, O.IMT_TN, out=gw, accumulate=True, a_colsum=gb, force_general=variant)
        assert_close(gw, gw0 + dyf.t() @ xf, tol, "TN ragged K variant %d" % variant)
        assert_close(gb, gb0 + dyf.sum(0), tol, "fused bias gradient, ragged K variant %d" % variant)


def test_gemm_bad_args(cuda):
    from imagetranslate_amd import hip_ops as O
    from imagetranslate_amd._lib import ImtError
    A = torch.zeros((16, 12), device=cuda)  # K=12 ok for f32 (mult of 4) but lda fine; bf16 needs mult of 8
    with pytest.raises(ImtError):
        O.gemm(A.bfloat16(), A.bfloat16(), O.IMT_NT)
    # empty problem is a no-op
    out = O.gemm(torch.zeros((0, 16), device=cuda), torch.zeros((8, 16), device=cuda), O.IMT_NT)
    assert out.shape == (0, 8)


# ------------------------------------------------------------------------------------------------ row ops
@pytest.mark.parametrize("d", [128, 512, 768])
@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm(cuda, dtype, d):
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(4)
    rows = 333
    x, xf = _mk((rows, d), dtype, cuda, 2.0, g)
    gamma, gf = _mk((d,), dtype, cuda, 1.0, g)
    beta, bf = _mk((d,), dtype, cuda, 1.0, g)
    dy, dyf = _mk((rows, d), dtype, cuda, 1.0, g)
    xr = xf.clone().requires_grad_(True)
    gr = gf.clone().requires_grad_(True)
    br = bf.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (d,), gr, br, eps=1e-12)
    ref.backward(dyf)
    tol = 1e-5 if dtype == torch.float32 else 1.5e-2
    y, mean, rstd = O.layernorm_fwd(x, gamma, beta, 1e-12)
    assert_close(y, ref, tol, "ln fwd")
    assert_close(mean, xf.mean(-1), 1e-5, "ln mean")
    dgamma = torch.zeros(d, device=cuda)
    dbeta = torch.zeros(d, device=cuda)
    dx = O.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta)
    assert_close(dx, xr.grad, tol * 2, "ln dx")
    assert_close(dgamma, gr.grad, 1e-4 if dtype == torch.float32 else 1e-2, "ln dgamma")
    assert_close(dbeta, br.grad, 1e-4 if dtype == torch.float32 else 1e-2, "ln dbeta")
    # accumulation semantics: a second call doubles the parameter grads
    O.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta)
    assert_close(dgamma, 2 * gr.grad, 1e-4 if dtype == torch.float32 else 1e-2, "ln dgamma accum")


@pytest.mark.parametrize("dtype", DTYPES)
def test_embedding(cuda, dtype):
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(5)
    V, P, NT, d, B, S = 1000, 512, 2, 128, 8, 32
    word, wf = _mk((V, d), dtype, cuda, 1.0, g)
    pos, pf = _mk((P, d), dtype, cuda, 1.0, g)
    typ, tf = _mk((NT, d), dtype, cuda, 1.0, g)
    ids = torch.randint(0, V, (B, S), generator=g)
    ids[:, -5:] = 0  # pads
    types = torch.randint(0, NT, (B, 1), generator=g).expand(B, S).contiguous()
    pos_ids = torch.randint(0, P, (B, S), generator=g)
    for use_pos in (False, True):
        pid = pos_ids if use_pos else torch.arange(S).expand(B, S)
        ref = wf[ids] + pf[pid] + tf[types]
        out = O.embed_fwd(ids.to(cuda), pos_ids.to(cuda) if use_pos else None, types.to(cuda), word, pos, typ, S)
        assert_close(out.view(B, S, d), ref, 1e-6 if dtype == torch.float32 else 1e-2, "embed fwd")
        dsum, dsf = _mk((B * S, d), dtype, cuda, 1.0, g)
        dw = torch.zeros((V, d), device=cuda); dp = torch.zeros((P, d), device=cuda); dtt = torch.zeros((NT, d), device=cuda)
        O.embed_bwd(ids.to(cuda), pos_ids.to(cuda) if use_pos else None, types.to(cuda), dsum, dw, dp, dtt, S, 0)
        rw = torch.zeros(V, d).index_add_(0, ids.view(-1), dsf); rw[0] = 0  # padding_idx row: no grad
        rp = torch.zeros(P, d).index_add_(0, pid.reshape(-1), dsf)
        rt = torch.zeros(NT, d).index_add_(0, types.view(-1), dsf)
        assert_close(dw, rw, 1e-5, "embed dword")
        assert_close(dp, rp, 1e-5, "embed dpos")
        assert_close(dtt, rt, 1e-5, "embed dtype")


@pytest.mark.parametrize("dtype", DTYPES)
def test_colsum_gather_scatter_mix_cast(cuda, dtype):
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(6)
    X, Xf = _mk((777, 264), dtype, cuda, 1.0, g)
    out = torch.zeros(264, device=cuda)
    O.colsum(X, out)
    assert_close(out, Xf.sum(0), 1e-5, "colsum")
    idx = torch.randperm(777, generator=g)[:300].sort().values.int()
    sel = O.gather_rows(X, idx.to(cuda))
    assert torch.equal(sel.float().cpu(), Xf[idx.long()])
    dx = torch.zeros_like(X)
    O.scatter_rows(sel, idx.to(cuda), dx)
    ref = torch.zeros_like(Xf); ref[idx.long()] = Xf[idx.long()]
    assert torch.equal(dx.float().cpu(), ref)
    a, af = _mk((50, 128), dtype, cuda, 1.0, g); b, bf = _mk((50, 128), dtype, cuda, 1.0, g)
    gate, gf = _mk((128,), dtype, cuda, 1.0, g)
    s = torch.sigmoid(gf + 1e-7)
    assert_close(O.gated_mix(a, b, gate), s * af + (1 - s) * bf, 1e-5 if dtype == torch.float32 else 1e-2, "gated mix")
    src = torch.randn(1003, generator=g)
    dst = torch.empty(1003, device=cuda, dtype=torch.bfloat16)
    O.cast_f32_to_bf16(src.to(cuda), dst)
    assert torch.equal(dst.cpu(), src.bfloat16())


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, B, H, Tq, Tk, dh, mask):
    """HF BertSelfAttention math; mask: bool [B,Tq,Tk] (True = attend)."""
    qh = q.view(B, Tq, H, dh).permute(0, 2, 1, 3)
    kh = k.view(B, Tk, H, dh).permute(0, 2, 1, 3)
    vh = v.view(B, Tk, H, dh).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(dh) + ((1.0 - mask.float()) * -10000.0)[:, None]
    p = F.softmax(s, dim=-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B * Tq, H * dh)


@pytest.mark.parametrize("case", ["enc_keymask", "dec_causal_qmask", "cross_49", "mask3d", "long", "mask3d_long", "causal_long"])
@pytest.mark.parametrize("dh", [32, 64])
@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_fwd_bwd(cuda, dtype, dh, case):
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(7)
    B, H = 3, 4
    Tq, Tk = {"enc_keymask": (32, 32), "dec_causal_qmask": (127, 127), "cross_49": (31, 49), "mask3d": (20, 70),
              "long": (130, 200), "mask3d_long": (150, 256), "causal_long": (256, 256)}[case]
    d = H * dh
    # q/k/v live in one wider buffer (fused QKV layout) for the self-attention cases
    q, qf = _mk((B * Tq, d), dtype, cuda, 1.0, g)
    k, kf = _mk((B * Tk, d), dtype, cuda, 1.0, g)
    v, vf = _mk((B * Tk, d), dtype, cuda, 1.0, g)
    do, dof = _mk((B * Tq, d), dtype, cuda, 1.0, g)
    key_mask = query_mask = mask3d = None
    causal = False
    if case in ("enc_keymask", "cross_49", "long"):
        lens = torch.randint(Tk // 2, Tk + 1, (B,), generator=g)
        key_mask = (torch.arange(Tk)[None] < lens[:, None])
        mask = key_mask[:, None, :].expand(B, Tq, Tk)
    elif case in ("dec_causal_qmask", "causal_long"):
        lens = torch.randint(Tq // 2, Tq + 1, (B,), generator=g)
        query_mask = (torch.arange(Tq)[None] < lens[:, None])
        causal = True
        mask = torch.tril(torch.ones(Tq, Tk, dtype=torch.bool))[None] & query_mask[:, :, None]
    else:
        mask3d = torch.rand((B, Tq, Tk), generator=g) > 0.3
        mask = mask3d
    # Fully masked (pad) query rows see a uniform -10000 shift: their softmax is ill-conditioned in fp32
    # (ulp(1e4) ~ 1e-3) and the model discards them (src/seq2seq.py:175-177), so they are excluded from the
    # comparison and carry no upstream gradient -- exactly as in the real model.
    valid_q = torch.ones(B * Tq, dtype=torch.bool) if query_mask is None else query_mask.reshape(-1)
    dof = dof * valid_q[:, None]
    do = (do.float() * valid_q[:, None].to(cuda)).to(dtype)
    u8 = lambda m: None if m is None else m.to(torch.uint8).contiguous().to(cuda)
    o, lse = O.attention_fwd(q, k, v, B, H, Tq, Tk, dh, key_mask=u8(key_mask), query_mask=u8(query_mask),
                             mask3d=u8(mask3d), causal=causal)
    qr, kr, vr = [t.clone().requires_grad_(True) for t in (qf, kf, vf)]
    ref = _attn_ref(qr, kr, vr, B, H, Tq, Tk, dh, mask)
    ref.backward(dof)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert_close(o.cpu()[valid_q], ref[valid_q], tol, "attn fwd " + case)
    dq, dk, dv = O.attention_bwd(do, q, k, v, o, lse, B, H, Tq, Tk, dh, key_mask=u8(key_mask), query_mask=u8(query_mask),
                                 mask3d=u8(mask3d), causal=causal)
    tolb = 5e-5 if dtype == torch.float32 else 3e-2
    assert_close(dq.cpu()[valid_q], qr.grad[valid_q], tolb, "attn dq " + case)
    assert_close(dk, kr.grad, tolb, "attn dk " + case)
    assert_close(dv, vr.grad, tolb, "attn dv " + case)


@pytest.mark.parametrize("dh,Tq,Tk,causal", [(64, 128, 128, False), (64, 127, 127, True), (32, 100, 128, False), (64, 128, 49, False),
                                             (64, 256, 256, False), (64, 255, 255, True), (64, 128, 256, False), (32, 200, 130, False),
                                             (64, 130, 200, False), (32, 256, 256, True)])
def test_attention_bwd_fused_matches_two_kernel_path(cuda, dh, Tq, Tk, causal):
    """bf16: the single fused backward kernel (Tq / Tk <= 128) and its 256-key sibling (129 .. 256: the MASS shapes of BASELINE
    configs[4]) against the dQ + dK/dV kernel pair on the same inputs, with key padding, query mask, dropout (same
    counter-based mask) and strided q|k|v views; the dispatch is asserted through the profiler kinds."""
    import os
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(9)
    B, H = 4, 3
    d = H * dh
    qkv = (torch.randn(B * Tq, 3 * d, generator=g) * 0.8).bfloat16().cuda()
    kv = qkv if Tq == Tk else (torch.randn(B * Tk, 3 * d, generator=g) * 0.8).bfloat16().cuda()
    q, k, v = qkv[:, :d], kv[:, d:2 * d], kv[:, 2 * d:]
    klen = torch.randint(Tk // 2, Tk + 1, (B,), generator=g)
    kmask = (torch.arange(Tk)[None] < klen[:, None]).to(torch.uint8).cuda()
    qmask = (torch.rand(B, Tq, generator=g) > 0.1).to(torch.uint8).cuda() if causal else None
    kw = dict(key_mask=kmask, query_mask=qmask, causal=causal, dropout_p=0.1, dropout_seed=4242)
    o, lse = O.attention_fwd(q, k, v, B, H, Tq, Tk, dh, **kw)
    do = torch.randn(B * Tq, d, generator=g).bfloat16().cuda()
    from imagetranslate_amd import _lib as L
    lib = L.load()
    res, kinds = {}, {}
    for name, env in (("fused", None), ("pair", "1")):
        if env is None:
            os.environ.pop("IMT_ATTN_NO_FUSED_BWD", None)
        else:
            os.environ["IMT_ATTN_NO_FUSED_BWD"] = env
        try:
            torch.cuda.synchronize()
            lib.imt_prof_enable(1)
            res[name] = [t.float().cpu() for t in O.attention_bwd(do, q, k, v, o, lse, B, H, Tq, Tk, dh, **kw)]
            torch.cuda.synchronize()
            rows = (L.ProfRow * 64)()
            kinds[name] = {rows[i].kind.decode() for i in range(lib.imt_prof_report(rows, 64))}
        finally:
            lib.imt_prof_enable(0)
            os.environ.pop("IMT_ATTN_NO_FUSED_BWD", None)
    big = max(Tq, Tk) > 128
    assert ("attn_bwd_fused256_bf16" if big else "attn_bwd_fused_bf16") in kinds["fused"], kinds
    assert {"attn_bwd_dq_bf16", "attn_bwd_dkdv_bf16"} <= kinds["pair"], kinds
    for what, a, b in zip(("dQ", "dK", "dV"), res["fused"], res["pair"]):
        assert_close(a, b, 1e-2, "fused vs two-kernel " + what)
    assert torch.equal(res["fused"][2], res["pair"][2]), "dV follows the same operation order in both paths"
    # same inputs, same launch: bit-identical (nothing is summed across workgroups)
    again = [t.float().cpu() for t in O.attention_bwd(do, q, k, v, o, lse, B, H, Tq, Tk, dh, **kw)]
    for a, b in zip(again, res["fused"]):
        assert torch.equal(a, b)


def test_attention_fused_qkv_views_and_dropout(cuda):
    """Strided q/k/v views into one [N,3d] buffer; dropout: deterministic, backward consistent with forward
    (finite-difference-free check: dropout with p -> compare against reference using the recovered mask)."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(8)
    B, H, T, dh = 2, 4, 48, 32
    d = H * dh
    qkv = torch.randn((B * T, 3 * d), generator=g).to(cuda)
    q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
    o, lse = O.attention_fwd(q, k, v, B, H, T, T, dh, causal=True)
    mask = torch.tril(torch.ones(T, T, dtype=torch.bool))[None].expand(B, T, T)
    ref = _attn_ref(q.cpu().contiguous(), k.cpu().contiguous(), v.cpu().contiguous(), B, H, T, T, dh, mask)
    assert_close(o, ref, 2e-5, "fused-qkv views")
    # dropout determinism + expected keep fraction via V = identity-like probe
    o1, _ = O.attention_fwd(q, k, v, B, H, T, T, dh, causal=True, dropout_p=0.5, dropout_seed=99)
    o2, _ = O.attention_fwd(q, k, v, B, H, T, T, dh, causal=True, dropout_p=0.5, dropout_seed=99)
    o3, _ = O.attention_fwd(q, k, v, B, H, T, T, dh, causal=True, dropout_p=0.5, dropout_seed=100)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    # backward with dropout: compare to autograd through an explicit-mask reference.  Recover the mask by
    # running the kernel with V = one-hot columns (P_drop rows appear directly).
    Tk = T
    eye = torch.zeros((B * Tk, d))
    for h in range(H):
        for j in range(min(Tk, dh)):
            eye.view(B, Tk, H, dh)[:, j, h, j] = 1.0
    pd, _ = O.attention_fwd(q, k, eye.to(cuda), B, H, T, Tk, dh, causal=True, dropout_p=0.5, dropout_seed=99)
    pd = pd.cpu().view(B, T, H, dh).permute(0, 2, 1, 3)[..., :min(Tk, dh)]  # P_drop[b,h,i,j<dh]
    qf, kf = q.cpu().contiguous(), k.cpu().contiguous()
    s = (qf.view(B, T, H, dh).permute(0, 2, 1, 3) @ kf.view(B, T, H, dh).permute(0, 2, 3, 1)) / math.sqrt(dh)
    s = s + ((1.0 - mask.float()) * -10000.0)[:, None]
    p = F.softmax(s, -1)[..., :min(Tk, dh)]
    keep = pd != 0
    assert_close(pd[keep], (p / 0.5)[keep], 1e-4, "dropped probs")
    visible = p > 1e-6
    frac = keep[visible].float().mean().item()
    assert abs(frac - 0.5) < 0.05, frac


# ------------------------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("dtype", DTYPES)
def test_log_softmax_and_smoothed_nll(cuda, dtype):
    from imagetranslate_amd import hip_ops as O
    from oracle.reference_model import SmoothedNLLLoss
    g = torch.Generator().manual_seed(9)
    N, V = 37, 1000
    z, zf = _mk((N, V), dtype, cuda, 3.0, g)
    tgt = torch.randint(1, V, (N,), generator=g)
    tgt[::7] = 0  # ignored rows
    zr = zf.clone().requires_grad_(True)
    lp_ref = F.log_softmax(zr, dim=-1)
    loss_ref = SmoothedNLLLoss(ignore_index=0)(lp_ref, tgt)
    loss_ref.mean().backward()
    lp, lse = O.log_softmax_fwd(z)
    assert_close(lp, lp_ref, 1e-6, "log_softmax")
    loss = O.smoothed_nll_fwd(lp, tgt.to(cuda), 0.1, 0)
    assert loss.shape == (N, 1)
    assert_close(loss, loss_ref, 1e-5, "smoothed nll")
    dloss = torch.full((N, 1), 1.0 / N, device=cuda)
    dlp = O.smoothed_nll_bwd(dloss, tgt.to(cuda), V, 0.1, 0)
    dz = O.log_softmax_bwd(dlp, lp, torch.float32)
    assert_close(dz, zr.grad, 1e-4, "dlogits (2-kernel path)")
    # fused
    z2 = z.clone()
    lrows = O.xent_fused_fwd_bwd(z2, tgt.to(cuda), 0.1, 0, 1.0 / N)
    assert_close(lrows.view(N, 1), loss_ref, 1e-5 if dtype == torch.float32 else 1e-4, "fused loss")
    assert_close(z2, zr.grad, 1e-4 if dtype == torch.float32 else 1e-2, "fused dlogits")


def test_loss_known_answer(cuda):
    """Known-answer vector produced by the reference's own src/loss.py (tests/golden/loss_kat.json)."""
    import json
    import os
    from imagetranslate_amd import hip_ops as O
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "loss_kat.json")))
    for case in kat["cases"]:
        logits = torch.tensor(case["logits"], dtype=torch.float32)
        tgt = torch.tensor(case["target"], dtype=torch.long)
        lp, _ = O.log_softmax_fwd(logits.to(cuda))
        loss = O.smoothed_nll_fwd(lp, tgt.to(cuda), case["epsilon"], case["ignore_index"])
        assert_close(loss.view(-1), torch.tensor(case["loss"]), 1e-6, "loss KAT")
        z = logits.to(cuda).clone()
        n = len(case["target"])
        O.xent_fused_fwd_bwd(z, tgt.to(cuda), case["epsilon"], case["ignore_index"], 1.0 / n)
        assert_close(z, torch.tensor(case["dlogits_mean"]), 1e-5, "loss KAT grad")


# ------------------------------------------------------------------------------------------------ optimizer
def test_clip_adam_matches_torch(cuda):
    from imagetranslate_amd import hip_ops as O
    from oracle.reference_model import AdamInverseSqrtWithWarmup
    g = torch.Generator().manual_seed(10)
    n = 10007
    p0 = torch.randn(n, generator=g)
    pr = torch.nn.Parameter(p0.clone())
    opt = AdamInverseSqrtWithWarmup([pr], lr=1e-3, betas=(0.9, 0.98), warmup_updates=3)
    p = torch.zeros(n + 1, device=cuda)[:n]  # (keeps 16-B alignment: offset 0)
    p.copy_(p0)
    m = torch.zeros(n, device=cuda); v = torch.zeros(n, device=cuda)
    pb = torch.empty(n, device=cuda, dtype=torch.bfloat16)
    from oracle.reference_model import inverse_sqrt_lr
    lr = 1e-7
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (5.0 if step % 2 else 0.001)
        pr.grad = grad.clone()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        gdev = grad.to(cuda)
        ss = torch.zeros(1, device=cuda)
        O.sumsq(gdev, ss)
        assert_close(ss, (grad.double() ** 2).sum().float().view(1), 1e-5, "sumsq")
        O.clip_adam(p, gdev, m, v, pb, ss, 1.0, 1.0, lr, 0.9, 0.98, 1e-8, step, zero_grad=True)
        lr = inverse_sqrt_lr(step, 1e-3, 3)
        assert_close(p, pr.detach(), 1e-5, "adam step %d" % step)
        assert float(gdev.abs().max()) == 0.0
        assert torch.equal(pb.cpu(), p.cpu().bfloat16())


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_splitk_slab_mode(cuda, dtype, layout):
    """IMT_AUX_SPLITK_WS: K-ranges of 256-tile workgroups into fp32 slabs + one reduce launch == the plain product, with a
    ragged K tail, alpha_dev, accumulate, ragged M / N and bf16 or fp32 output (exact on integer data)."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(77)
    bk = 64 if dtype == torch.bfloat16 else 32
    for (M, N, K, splits) in [(300, 264, 16 * bk, 4), (520, 128, 9 * bk + 24, 2), (256, 256, 40 * bk, 8)]:
        A = torch.randint(-2, 3, (M, K), generator=g).float()
        B = torch.randint(-2, 3, (N, K), generator=g).float()
        ref = A @ B.t()
        da = A.to(dtype).to(cuda)
        db = (B if layout == O.IMT_NT else B.t().contiguous()).to(dtype).to(cuda)
        slabs = torch.empty(splits * M, N, device=cuda, dtype=torch.float32)
        out = torch.empty(M, N, device=cuda, dtype=torch.float32)
        O.gemm(da, db, layout, out=out, aux=slabs, aux_mode=O.IMT_AUX_SPLITK_WS, split_k=splits)
        assert torch.equal(out.cpu(), ref), (M, N, K, splits)
        c0 = torch.randint(-4, 5, (M, N), generator=g).float()
        out = c0.clone().to(cuda)
        scale = torch.tensor([0.5], device=cuda)
        O.gemm(da, db, layout, out=out, aux=slabs, aux_mode=O.IMT_AUX_SPLITK_WS, split_k=splits, accumulate=True, alpha_dev=scale)
        assert torch.equal(out.cpu(), c0 + 0.5 * ref), "accumulate + alpha_dev"
        outT = torch.empty(M, N, device=cuda, dtype=dtype)
        O.gemm(da, db, layout, out=outT, aux=slabs, aux_mode=O.IMT_AUX_SPLITK_WS, split_k=splits)
        assert_close(outT.float(), ref, 1e-6 if dtype == torch.float32 else 8e-3, "compute-dtype output")


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_bwd_partial_sums(cuda, dtype):
    """imt_layernorm_bwd with a partial-sum workspace + imt_ln_partial_reduce == the straight-atomics form: same dx,
    dgamma / dbeta equal up to summation order, a skipped slot (negative offset) left alone, accumulation onto old values."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(5)
    rows, d = 1000, 384
    x, _ = _mk((rows, d), dtype, cuda, 1.0, g)
    dy, _ = _mk((rows, d), dtype, cuda, 1.0, g)
    gamma, _ = _mk((d,), dtype, cuda, 1.0, g)
    beta, _ = _mk((d,), dtype, cuda, 1.0, g)
    y, mean, rstd = O.layernorm_fwd(x, gamma, beta)
    grads_ref = torch.randn(4 * d, generator=g).to(cuda)
    grads = grads_ref.clone()
    init = grads_ref.clone()
    dx_ref = O.layernorm_bwd(dy, x, gamma, mean, rstd, grads_ref[d:2 * d], grads_ref[3 * d:4 * d])
    parts = torch.zeros(3, O.LN_PARTIAL_COPIES, 2, d, device=cuda)
    parts[1].fill_(7.0)  # a slot the reduce must skip
    dx = O.layernorm_bwd(dy, x, gamma, mean, rstd, grads[d:2 * d], grads[3 * d:4 * d], partial_ws=parts[2])
    assert torch.equal(dx, dx_ref)
    before = grads.clone()
    assert torch.equal(before, init), "the gradients must be untouched until the reduce"
    O.ln_partial_reduce(parts, grads, [-1, -1, d], [-1, -1, 3 * d])
    tol = 2e-5 if dtype == torch.float32 else 2e-3
    assert_close(grads[d:2 * d], grads_ref[d:2 * d], tol, "dgamma through partial sums")
    assert_close(grads[3 * d:4 * d], grads_ref[3 * d:4 * d], tol, "dbeta through partial sums")
    assert torch.equal(grads[:d], before[:d]) and torch.equal(grads[2 * d:3 * d], before[2 * d:3 * d])


# ------------------------------------------------------------------------------------------------ fused dense + LayerNorm
@pytest.mark.parametrize("shape", [(8128, 512, 512), (300, 128, 128), (33, 256, 64), (1000, 384, 2048), (8192, 512, 2048), (5, 512, 128), (70, 512, 64)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_bias_residual_ln_matches_separate_kernels(cuda, dtype, shape):
    """imt_gemm_bias_residual_ln == imt_gemm(bias, dropout, residual epilogue) followed by imt_layernorm_fwd: the pre-LN
    matrix equal up to the round-off of a different K-step order (same products, same epilogue arithmetic, same dropout
    element index -- the dropped positions are identical), LayerNorm output / statistics likewise; and both against torch."""
    from imagetranslate_amd import hip_ops as O
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    x, xr = _mk((M, K), dtype, cuda, gen=g)
    w, wr = _mk((N, K), dtype, cuda, scale=1.0 / math.sqrt(K), gen=g)
    b, br = _mk((N,), dtype, cuda, scale=0.2, gen=g)
    res, rr = _mk((M, N), dtype, cuda, gen=g)
    gam, gr = _mk((N,), dtype, cuda, gen=g)
    bet, ber = _mk((N,), dtype, cuda, scale=0.3, gen=g)
    for p, with_res, with_bias in ((0.0, True, True), (0.1, True, True), (0.0, False, False)):
        pre2 = O.gemm(x, w, O.IMT_NT, bias=b if with_bias else None, resid=res if with_res else None, dropout_p=p, dropout_seed=77)
        out2, mean2, rstd2 = O.layernorm_fwd(pre2, gam, bet, eps=1e-12)
        out, pre, mean, rstd = O.gemm_bias_residual_ln(x, w, b if with_bias else None, res if with_res else None, gam, bet,
                                                       eps=1e-12, dropout_p=p, dropout_seed=77)
        # same products, same epilogue arithmetic, same dropout element index; only the order of the K steps differs per
        # workgroup (K-order rotation) -- round-off of an fp32 sum / one bf16 rounding step
        assert_close(pre, pre2.float().cpu(), 1e-5 if dtype == torch.float32 else 1.6e-2, "pre-LN vs imt_gemm (p=%g)" % p)
        if p > 0:
            # the dropped positions (pre-LN == residual exactly) are the same in both paths, up to the rare element whose
            # product is below half an ulp of the residual in one summation order and not in the other
            da, db = (pre == res if with_res else pre == 0), (pre2 == res if with_res else pre2 == 0)
            rate_tol = 4.0 * math.sqrt(p * (1 - p) / da.numel()) + 0.002
            assert abs(float(da.float().mean()) - p) < rate_tol and float((da != db).float().mean()) < 1e-4, "dropout masks differ"
        tol = 1e-5 if dtype == torch.float32 else 3.2e-2  # bf16: a rounding step of the pre-LN value and one of the output
        assert_close(out, out2.float().cpu(), tol, "fused LN output vs separate kernels")
        stol = 1e-5 if dtype == torch.float32 else 2e-3  # bf16: statistics of pre-LN values that differ by rounding steps
        assert_close(mean, mean2.cpu(), stol, "mean")
        assert_close(rstd, rstd2.cpu(), stol, "rstd")
        if p == 0.0:
            ref_pre = xr @ wr.t() + (br if with_bias else 0) + (rr if with_res else 0)
            ref = F.layer_norm(ref_pre.to(dtype).float(), (N,), gr, ber, eps=1e-12)
            assert_close(out, ref, TOL[dtype], "fused LN output vs torch")
    # no pre-LN output requested (inference): same LayerNorm output
    out3, pre3, _, _ = O.gemm_bias_residual_ln(x, w, b, res, gam, bet, want_pre_ln=False)
    assert pre3 is None
    out4, _, _, _ = O.gemm_bias_residual_ln(x, w, b, res, gam, bet)
    assert torch.equal(out3, out4)


def test_gemm_bias_residual_ln_bad_args(cuda):
    from imagetranslate_amd import hip_ops as O
    from imagetranslate_amd._lib import ImtError
    x = torch.zeros(64, 64, device=cuda)
    g = torch.ones(640, device=cuda)
    with pytest.raises(ImtError):  # N = 640 is not supported by the row-complete tile
        O.gemm_bias_residual_ln(x, torch.zeros(640, 64, device=cuda), None, None, g, g)
    with pytest.raises(ImtError):  # K = 8 fp32 = 32 bytes: not a whole 128-byte K tile
        O.gemm_bias_residual_ln(x[:, :8].contiguous(), torch.zeros(128, 8, device=cuda), None, None, g[:128], g[:128])


@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_ln_and_add_ln_fused_forwards(cuda, dtype):
    """imt_embed_ln_fwd == imt_embed_fwd + imt_layernorm_fwd and imt_add_layernorm_fwd == (x + resid) + imt_layernorm_fwd,
    bit for bit (same arithmetic, the sum rounded to T before LayerNorm like the stored sum of the two-launch path)."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(12)
    V, P, TY, d, B, T = 300, 64, 2, 384, 5, 37
    word, _ = _mk((V, d), dtype, cuda, gen=g)
    pos, _ = _mk((P, d), dtype, cuda, gen=g)
    typ, _ = _mk((TY, d), dtype, cuda, gen=g)
    gam, gr = _mk((d,), dtype, cuda, gen=g)
    bet, br = _mk((d,), dtype, cuda, scale=0.2, gen=g)
    ids = torch.randint(0, V, (B, T), generator=g).to(cuda)
    tids = torch.randint(0, TY, (B, T), generator=g).to(cuda)
    pids = torch.randint(0, P, (B, T), generator=g).to(cuda)
    for pos_ids in (None, pids):
        for p in (0.0, 0.1):
            s2 = O.embed_fwd(ids, pos_ids, tids, word, pos, typ, T)
            y2, m2, r2 = O.layernorm_fwd(s2, gam, bet, dropout_p=p, dropout_seed=9)
            y, ssum, m, r = O.embed_ln_fwd(ids, pos_ids, tids, word, pos, typ, gam, bet, T, dropout_p=p, dropout_seed=9)
            assert torch.equal(ssum, s2) and torch.equal(y, y2) and torch.equal(m, m2) and torch.equal(r, r2)
    ref = F.layer_norm((word.float()[ids] + pos.float()[pids] + typ.float()[tids]).to(dtype).float().cpu().view(-1, d), (d,), gr, br, eps=1e-12)
    y, _, _, _ = O.embed_ln_fwd(ids, pids, tids, word, pos, typ, gam, bet, T)
    assert_close(y, ref, TOL[dtype], "embed + LN vs torch")
    x, _ = _mk((B * T, d), dtype, cuda, gen=g)
    res, _ = _mk((B * T, d), dtype, cuda, gen=g)
    y, ssum, m, r = O.add_layernorm_fwd(x, res, gam, bet)
    s2 = (x.float() + res.float()).to(dtype)
    y2, m2, r2 = O.layernorm_fwd(s2, gam, bet)
    assert torch.equal(ssum, s2) and torch.equal(y, y2) and torch.equal(m, m2) and torch.equal(r, r2)


# ------------------------------------------------------------------------------------------------ imt_gemm + in-launch LayerNorm
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(8192, 512, 512), (8128, 512, 2048), (1000, 256, 128), (8192, 1024, 256), (300, 384, 64)])
def test_gemm_with_layernorm_of_finished_rows(cuda, dtype, shape):
    """imt_gemm(ln_out=...): C keeps dropout(x W^T + b) + resid, ln_out = LayerNorm(C) -- in the GEMM's own launch for
    one-tile-per-workgroup launches of the persistent kernel (row-block tickets, write-through C, the last column tile of
    a row block normalises it), behind it otherwise.  Against the two-launch path (same C bit for bit; LayerNorm within
    round-off: the in-launch pass sums a row in a different lane order) and torch's LayerNorm of that C."""
    import imagetranslate_amd.hip_ops as O
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) / K ** 0.25).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.25).to(dtype).cuda()
    b, gam, bet = (torch.randn(N, generator=g).to(dtype).cuda() for _ in range(3))
    r = torch.randn(M, N, generator=g).to(dtype).cuda()
    tickets = torch.zeros((M + 127) // 128 + 1, dtype=torch.int32, device="cuda")
    for p in (0.0, 0.1):
        ref_c = O.gemm(x, w, O.IMT_NT, bias=b, resid=r, dropout_p=p, dropout_seed=11)
        ref_y, _, _ = O.layernorm_fwd(ref_c, gam, bet, eps=1e-12)
        c = torch.empty_like(ref_c)
        y = torch.empty_like(ref_c)
        mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
        for rep in range(2):  # twice: the tickets must come back to zero
            y.fill_(float("nan"))
            O.gemm(x, w, O.IMT_NT, out=c, bias=b, resid=r, dropout_p=p, dropout_seed=11,
                   ln=dict(gamma=gam, beta=bet, out=y, mean=mean, rstd=rstd, tickets=tickets, eps=1e-12))
            torch.cuda.synchronize()
            assert int(tickets.abs().sum()) == 0, "tickets must return to zero"
            assert torch.equal(c, ref_c), "the LayerNorm input is the plain epilogue's result"
            want = torch.nn.functional.layer_norm(ref_c.float(), (N,), gam.float(), bet.float(), 1e-12)
            tol = 1e-5 if dtype == torch.float32 else 1.2e-2
            assert_close(y.float(), want, tol, "in-launch LayerNorm vs torch")
            assert_close(y.float(), ref_y.float(), 1e-5 if dtype == torch.float32 else 8e-3, "in-launch LayerNorm vs the two-launch path")
            assert_close(mean, ref_c.float().mean(1), 1e-5 if dtype == torch.float32 else 1e-4, "row means")
            assert_close(rstd, 1.0 / torch.sqrt(ref_c.float().var(1, unbiased=False) + 1e-12), 1e-4, "row rstd")


# ------------------------------------------------------------------------------------------------ split-K for few output tiles
def _kinds(fn):
    from imagetranslate_amd import _lib as L
    lib = L.load()
    torch.cuda.synchronize()
    lib.imt_prof_enable(1)
    try:
        out = fn()
        torch.cuda.synchronize()
        rows = (L.ProfRow * 64)()
        kinds = {rows[i].kind.decode(): int(rows[i].launches) for i in range(lib.imt_prof_report(rows, 64))}
    finally:
        lib.imt_prof_enable(0)
    return out, kinds


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_small_m_split_k_with_epilogues(cuda, dtype, layout):
    """Few output tiles and a long K (decoding: batch x beam = 320 rows, src/seq_gen.py:164-194; captioning: 32 x 31 caption
    tokens, src/train_captioning.py:51-72): with a workspace, imt_gemm runs K ranges on the persistent kernel + one epilogue
    launch.  Every epilogue the runtime uses, against the unsplit product (summation order differs: tolerance) and the fp32
    reference; two runs bit-identical (fixed summation order, no atomics); dispatch asserted through the profiler kinds."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(31)
    bk = 64 if dtype == torch.bfloat16 else 32
    tol = 3e-5 if dtype == torch.float32 else 1.5e-2
    ws = O.splitk_workspace(cuda)
    for (M, N, nk, expect_split) in [(320, 512, 32, True), (992, 512, 32, True), (1000, 264, 24, True), (61, 128, 16, True), (1, 512, 32, True), (320, 200, 32, True),
                                     (320, 512, 8, False), (4224, 512, 32, False)]:
        K = nk * bk
        A, Af = _mk((M, K), dtype, cuda, 0.5, g)
        B, Bf = _mk((N, K) if layout == O.IMT_NT else (K, N), dtype, cuda, 0.2, g)
        bias, biasf = _mk((N,), dtype, cuda, 1.0, g)
        R, Rf = _mk((M, N), dtype, cuda, 1.0, g)
        base = Af @ (Bf.t() if layout == O.IMT_NT else Bf)
        out, kinds = _kinds(lambda: O.gemm(A, B, layout, bias=bias, splitk_ws=ws))
        assert ("gemm_splitk_epilogue" in kinds) == expect_split, (M, N, K, kinds)
        assert_close(out, base + biasf, tol, "split-K bias")
        if not expect_split:
            assert torch.equal(out, O.gemm(A, B, layout, bias=bias)), "without a split the workspace must change nothing"
            continue
        assert_close(out, O.gemm(A, B, layout, bias=bias), tol, "split vs unsplit")
        assert torch.equal(out, O.gemm(A, B, layout, bias=bias, splitk_ws=ws)), "fixed summation order"
        # GELU (pre-activation saved), GELU', dropout + residual, accumulate into fp32 and into the operand type
        aux = torch.empty((M, N), device=cuda, dtype=dtype)
        h = O.gemm(A, B, layout, bias=bias, aux=aux, aux_mode=O.IMT_AUX_GELU_FWD, splitk_ws=ws)
        assert_close(aux, base + biasf, tol, "split-K gelu aux")
        assert_close(h, F.gelu(base + biasf), tol, "split-K gelu")
        z = (base + biasf).to(dtype).float().requires_grad_(True)
        F.gelu(z).sum().backward()
        assert_close(O.gemm(A, B, layout, aux=aux, aux_mode=O.IMT_AUX_DGELU, splitk_ws=ws), base * z.grad, 2 * tol, "split-K dgelu")
        d1 = O.gemm(A, B, layout, bias=bias, resid=R, dropout_p=0.2, dropout_seed=99, splitk_ws=ws).float().cpu()
        d0 = O.gemm(A, B, layout, bias=bias, resid=R, dropout_p=0.2, dropout_seed=99).float().cpu()
        assert_close(d1, d0, tol, "split-K dropout + residual (same mask as the GEMM epilogue)")
        assert float(((d1 - Rf) == 0).float().mean()) == pytest.approx(0.2, abs=0.03)
        C0 = torch.randn((M, N), generator=g)
        C = C0.clone().to(cuda)
        O.gemm(A, B, layout, out=C, accumulate=True, alpha=0.5, splitk_ws=ws)
        assert_close(C, C0 + 0.5 * base, tol, "split-K alpha + accumulate f32")
        Ct, Ctf = _mk((M, N), dtype, cuda, 1.0, g)
        O.gemm(A, B, layout, out=Ct, accumulate=True, splitk_ws=ws)
        assert_close(Ct, Ctf + base, 2 * tol, "split-K accumulate")
        # strided C (the decode step writes q|k|v of the new position into a cache row) and strided A
        wide = torch.zeros((M, 3 * N), device=cuda, dtype=dtype)
        O.gemm(A, B, layout, out=wide[:, N:2 * N], bias=bias, splitk_ws=ws)
        assert_close(wide[:, N:2 * N], base + biasf, tol, "split-K strided C")
        assert float(wide[:, :N].abs().max()) == 0 and float(wide[:, 2 * N:].abs().max()) == 0
        # LayerNorm of the finished rows behind the split product
        gam, gamf = _mk((N,), dtype, cuda, 1.0, g)
        bet, betf = _mk((N,), dtype, cuda, 1.0, g)
        ln = dict(gamma=gam, beta=bet, out=torch.empty((M, N), device=cuda, dtype=dtype), mean=torch.empty(M, device=cuda),
                  rstd=torch.empty(M, device=cuda), tickets=torch.zeros((M + 127) // 128 + 1, dtype=torch.int32, device=cuda))
        pre = O.gemm(A, B, layout, bias=bias, resid=R, ln=ln, splitk_ws=ws)
        ref_ln = F.layer_norm(pre.float().cpu(), (N,), gamf, betf, 1e-12)
        assert_close(ln["out"], ref_ln, 4 * tol if dtype == torch.float32 else 3e-2, "split-K + LayerNorm")


def test_gemm_split_k_long_ragged_k(cuda):
    """dX through the vocabulary with few rows (captioning: 992 x 512 x 30000): the ragged tail + whole-tile body split of
    imt_gemm, the body on K ranges, scaled by a device scalar."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(32)
    M, N, K = 992, 512, 30000
    A, Af = _mk((M, K), torch.bfloat16, cuda, 0.05, g)
    B, Bf = _mk((K, N), torch.bfloat16, cuda, 0.2, g)
    gdev = torch.full((1,), 0.5, device=cuda)
    out, kinds = _kinds(lambda: O.gemm(A, B, O.IMT_NN, alpha_dev=gdev, splitk_ws=O.splitk_workspace(cuda)))
    assert "gemm_splitk_epilogue" in kinds, kinds
    assert_close(out, 0.5 * (Af @ Bf), 1.5e-2, "ragged K + split-K")


@pytest.mark.parametrize("T,H,causal,drop", [(128, 8, False, 0.1), (127, 8, True, 0.1), (100, 4, False, 0.0), (65, 2, True, 0.0), (128, 8, True, 0.0)])
def test_attention_with_fused_qkv_projection_is_bit_identical(cuda, T, H, causal, drop):
    """q|k|v projection + short self-attention in one launch (imt_attention_qkv_fwd) against imt_gemm + imt_attention_fwd on the
    same inputs: q|k|v, the context and the log-sum-exp must be BIT-identical (same k order per accumulator, same attention
    code), with key padding, a query mask, causal masking and dropout; rows of the next batch element that the 128-row tile
    touches (T < 128) must not leak into the result."""
    from imagetranslate_amd import hip_ops as O
    g = torch.Generator().manual_seed(41)
    B, dh = 5, 64
    d = H * dh
    x = (torch.randn(B * T, d, generator=g) * 0.7).bfloat16().cuda()
    w = (torch.randn(3 * d, d, generator=g) * 0.06).bfloat16().cuda()
    bias = (torch.randn(3 * d, generator=g) * 0.1).bfloat16().cuda()
    klen = torch.randint(T // 2, T + 1, (B,), generator=g)
    kmask = (torch.arange(T)[None] < klen[:, None]).to(torch.uint8).cuda()
    qmask = (torch.rand(B, T, generator=g) > 0.1).to(torch.uint8).cuda() if causal else None
    kw = dict(key_mask=kmask, query_mask=qmask, causal=causal, dropout_p=drop, dropout_seed=777)
    qkv_ref = O.gemm(x, w, O.IMT_NT, bias=bias)
    o_ref, lse_ref = O.attention_fwd(qkv_ref[:, :d], qkv_ref[:, d:2 * d], qkv_ref[:, 2 * d:], B, H, T, T, dh, **kw)
    (qkv, o, lse), kinds = _kinds(lambda: O.attention_qkv_fwd(x, w, bias, B, H, T, dh, **kw))
    assert "attn_qkv_fwd_bf16" in kinds, kinds
    assert torch.equal(qkv, qkv_ref), "q|k|v differ (max %g)" % float((qkv.float() - qkv_ref.float()).abs().max())
    assert torch.equal(o, o_ref), "context differs (max %g)" % float((o.float() - o_ref.float()).abs().max())
    assert torch.equal(lse, lse_ref)
    # without a bias
    qkv_nb, o_nb, _ = O.attention_qkv_fwd(x, w, None, B, H, T, dh, **kw)
    qr = O.gemm(x, w, O.IMT_NT)
    assert torch.equal(qkv_nb, qr)
    assert torch.equal(o_nb, O.attention_fwd(qr[:, :d], qr[:, d:2 * d], qr[:, 2 * d:], B, H, T, T, dh, **kw)[0])
