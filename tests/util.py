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


TOL = {torch.float32: 1e-4, torch.bfloat16: 2.5e-2}
