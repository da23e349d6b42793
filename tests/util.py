import torch


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b| (b = reference), computed in fp64 on CPU."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / denom


def assert_close(a, b, tol, what=""):
    assert a.shape == b.shape, "%s shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    assert torch.isfinite(a.detach().float()).all(), "%s has non-finite values" % what
    e = rel_err(a, b)
    assert e <= tol, "%s: rel err %.3e > tol %.1e" % (what, e, tol)
    return e


def truth_errors(a: torch.Tensor, truth: torch.Tensor):
    """(max |a-t| / max |t|,  max over entries with |t| >= 1 % of max |t| of |a-t| / |t|) against an fp64 ground truth."""
    a = a.detach().double().cpu()
    t = truth.detach().double().cpu()
    assert a.shape == t.shape, (tuple(a.shape), tuple(t.shape))
    m = float(t.abs().max())
    if m == 0.0:
        return float(a.abs().max()), 0.0
    d = (a - t).abs()
    big = t.abs() >= 1e-2 * m
    return float(d.max()) / m, float((d[big] / t.abs()[big]).max())


def fp64_truth_report(name, ours, oracle32, truth, tol, elem_tol, log=None):
    """north_star's bar, measured against an fp64 run of the oracle: the HIP fp32 result within ``tol`` of the truth in the
    max norm and within ``elem_tol`` element-wise on the entries above 1 % of the maximum; the fp32 oracle's own distance
    from the truth is printed beside it (and written to ``log``), so that a bar is evidenced, not asserted."""
    e_o, r_o = truth_errors(ours, truth)
    e_f, r_f = truth_errors(oracle32, truth)
    line = "%-62s hip-fp32 %.2e (elem %.2e) | oracle-fp32 %.2e (elem %.2e)" % (name, e_o, r_o, e_f, r_f)
    print(line)
    if log is not None:
        log.append(line)
    assert torch.isfinite(ours.detach().float()).all(), name
    assert e_o <= tol, "%s: %.3e > %.1e against the fp64 truth (fp32 oracle: %.3e)" % (name, e_o, tol, e_f)
    assert r_o <= elem_tol, "%s: element-wise %.3e > %.1e against the fp64 truth (fp32 oracle: %.3e)" % (name, r_o, elem_tol, r_f)
    return e_o, e_f


TOL = {torch.float32: 1e-4, torch.bfloat16: 2.5e-2}


# ------------------------------------------------------------------------------------------- beam-search fixtures
def beam_state_dict(sd, scale: float = 4.0, eos_bias: float = 5.0, eos: int = 4):
    """Toy weights for beam search: the N(0, 0.02) fixture weights give near-uniform next-token distributions, so
    matrices are scaled up and the EOS logit biased -- hypotheses then finish at different steps and the EOS /
    length-limit / PAD bookkeeping of src/seq_gen.py:193-227 is exercised."""
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if v.dim() > 1 and v.is_floating_point():
            v = v * scale
        if k.startswith("output_layer") and k.endswith("layer.bias"):
            v[eos] = eos_bias
        out[k] = v
    return out


def beam_inputs():
    B, S = 6, 12
    g = torch.Generator().manual_seed(1)
    src = torch.randint(6, 1000, (B, S), generator=g)
    lens = torch.tensor([12, 10, 8, 12, 5, 7])
    mask = torch.arange(S)[None, :] < lens[:, None]
    src[~mask] = 0
    src[:, 0] = 5
    for b in range(B):
        src[b, lens[b] - 1] = 4
    return dict(src_inputs=src, src_sizes=lens, first_tokens=torch.full((B,), 5, dtype=torch.long), src_mask=mask,
                src_langs=torch.zeros(B, dtype=torch.long), tgt_langs=torch.ones(B, dtype=torch.long))


def caption_beam_inputs():
    B = 5
    g = torch.Generator().manual_seed(3)
    return dict(images=torch.randn(B, 49, 64, generator=g), first_tokens=torch.full((B,), 5, dtype=torch.long),
                tgt_langs=torch.ones(B, dtype=torch.long))
