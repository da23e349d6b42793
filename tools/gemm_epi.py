#!/usr/bin/env python3
"""Device time of imt_gemm WITH the epilogues the train step uses, per kernel variant, cycling over NSET operand sets
so that operands are not MALL-resident from the previous launch (as in the real step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O
from imagetranslate_amd import _lib as L

NSET = 6
dt = torch.bfloat16

def dev_time(fns, reps=3):
    lib = L.load()
    for f in fns: f()
    torch.cuda.synchronize()
    lib.imt_prof_enable(1)
    for _ in range(reps):
        for f in fns: f()
    torch.cuda.synchronize()
    rows = (L.ProfRow * 64)()
    n = lib.imt_prof_report(rows, 64)
    lib.imt_prof_enable(0)
    return sum(rows[i].total_ms for i in range(n)) * 1e3 / (reps * len(fns))

def case(name, lay, M, N, K, bias=False, aux=None, drop=0.0, resid=False):
    sets = []
    for s in range(NSET):
        A = torch.randn(M, K, device="cuda").to(dt)
        B = (torch.randn(N, K, device="cuda") if lay == O.IMT_NT else torch.randn(K, N, device="cuda")).to(dt)
        sets.append(dict(A=A, B=B, out=torch.empty(M, N, device="cuda", dtype=dt),
                         bias=torch.randn(N, device="cuda").to(dt) if bias else None,
                         aux=torch.randn(M, N, device="cuda").to(dt) if aux else None,
                         resid=torch.randn(M, N, device="cuda").to(dt) if resid else None))
    res = []
    for variant in (1, 3, 5):
        fns = [(lambda s=s: O.gemm(s["A"], s["B"], lay, out=s["out"], bias=s["bias"], aux=s["aux"], aux_mode=aux or O.IMT_AUX_NONE,
                                   dropout_p=drop, dropout_seed=7, resid=s["resid"], force_general=variant)) for s in sets]
        us = dev_time(fns)
        res.append("%6.1f us %4.0f TF" % (us, 2.0 * M * N * K / us / 1e6))
    print("%-34s dbuf %s | sbuf %s | ws %s" % (name, *res), flush=True)

if __name__ == "__main__":
    T, d, ff = 8192, 512, 2048
    case("ffn1 fwd plain        NT 2048x512", O.IMT_NT, T, ff, d)
    case("ffn1 fwd bias+GELU    NT 2048x512", O.IMT_NT, T, ff, d, bias=True, aux=O.IMT_AUX_GELU_FWD)
    case("ffn2 dx plain         NN 2048x512", O.IMT_NN, T, ff, d)
    case("ffn2 dx DGELU         NN 2048x512", O.IMT_NN, T, ff, d, aux=O.IMT_AUX_DGELU)
    case("attn-out plain        NT 512x512", O.IMT_NT, T, d, d)
    case("attn-out bias+drop+res NT 512x512", O.IMT_NT, T, d, d, bias=True, drop=0.1, resid=True)
    case("ffn2 fwd plain        NT 512x2048", O.IMT_NT, T, d, ff)
    case("ffn2 fwd bias+drop+res NT 512x2048", O.IMT_NT, T, d, ff, bias=True, drop=0.1, resid=True)
    case("qkv dx + resid        NN 512x1536", O.IMT_NN, T, d, 3 * d, resid=True)
