#!/usr/bin/env python3
"""Timing of the HBM-bound row kernels at C1 sizes (tools only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O

from imagetranslate_amd import _lib as L

def timeit(fn, reps=20):
    """pure device time per call: per-launch hipEvents recorded by the library around every kernel"""
    lib = L.load()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    lib.imt_prof_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    rows = (L.ProfRow * 64)()
    n = lib.imt_prof_report(rows, 64)
    lib.imt_prof_enable(0)
    return sum(rows[i].total_ms for i in range(n)) * 1e3 / reps

rows, d = 8192, 512
for dt in ((torch.bfloat16,) if __name__ == "__main__" else ()):
    x = torch.randn(rows, d, device="cuda").to(dt); dy = torch.randn(rows, d, device="cuda").to(dt)
    g = torch.ones(d, device="cuda").to(dt); b = torch.zeros(d, device="cuda").to(dt)
    y, mean, rstd = O.layernorm_fwd(x, g, b)
    dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
    es = 2
    t = timeit(lambda: O.layernorm_fwd(x, g, b)); print("ln_fwd            %7.1f us  %6.0f GB/s" % (t, 2 * rows * d * es / t / 1e3))
    t = timeit(lambda: O.layernorm_bwd(dy, x, g, mean, rstd, dg, db)); print("ln_bwd atomics    %7.1f us  %6.0f GB/s" % (t, 3 * rows * d * es / t / 1e3))
    t = timeit(lambda: O.layernorm_bwd(dy, x, g, mean, rstd, dg, db, want_dx_drop=True, dx_dropout_p=0.1, dx_dropout_seed=5)); print("ln_bwd +dropout   %7.1f us  %6.0f GB/s" % (t, 4 * rows * d * es / t / 1e3))
    out = torch.zeros(d, device="cuda")
    t = timeit(lambda: O.colsum(dy, out)); print("colsum            %7.1f us  %6.0f GB/s" % (t, rows * d * es / t / 1e3))
    q = torch.randn(rows, 3 * d, device="cuda").to(dt)
    B, H, T, dh = 64, 8, 128, 64
    t = timeit(lambda: O.attention_fwd(q[:, :d], q[:, d:2*d], q[:, 2*d:], B, H, T, T, dh)); print("attn_fwd          %7.1f us" % t)
    o, lse = O.attention_fwd(q[:, :d], q[:, d:2*d], q[:, 2*d:], B, H, T, T, dh)
    do = torch.randn(rows, d, device="cuda").to(dt)
    t = timeit(lambda: O.attention_bwd(do, q[:, :d], q[:, d:2*d], q[:, 2*d:], o, lse, B, H, T, T, dh)); print("attn_bwd          %7.1f us" % t)
    t = timeit(lambda: O.attention_fwd(q[:, :d], q[:, d:2*d], q[:, 2*d:], B, H, T, T, dh, causal=True)); print("attn_fwd causal   %7.1f us" % t)
    t = timeit(lambda: O.attention_fwd(q[:, :d], q[:, d:2*d], q[:, 2*d:], B, H, T, T, dh, dropout_p=0.1, dropout_seed=3)); print("attn_fwd dropout  %7.1f us" % t)
