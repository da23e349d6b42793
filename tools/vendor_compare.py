#!/usr/bin/env python3
"""Yardstick only (NOT used by the product): the vendor library (torch.matmul -> hipBLASLt) against imt_gemm on the C1 shapes with
HBM-cold operands (a rotating set of operand buffers), plain epilogue, and the names of the vendor kernels (their macro-tile and
wave layout are in the name).  GPU only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O
from torch.profiler import profile, ProfilerActivity


def sets(M, N, K, nt, n):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    out = []
    for _ in range(n):
        A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        B = (torch.randn((N, K) if nt else (K, N), device="cuda", generator=g) * 0.05).bfloat16()
        out.append((A, B, torch.empty(M, N, device="cuda", dtype=torch.bfloat16)))
    return out


def time_it(f, s, reps):
    for x in s[:2]:
        f(*x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        f(*s[r % len(s)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, M, N, K, nt in [("attn-out fwd", 8192, 512, 512, True), ("qkv fwd", 8192, 1536, 512, True), ("ffn1 fwd", 8192, 2048, 512, True),
                          ("ffn2 fwd", 8192, 512, 2048, True), ("vocab fwd", 8128, 30000, 512, True), ("ffn1 dx", 8192, 512, 2048, False),
                          ("ffn2 dx", 8192, 2048, 512, False), ("square", 8192, 2048, 2048, True)]:
    big = M * N > 1e8
    s = sets(M, N, K, nt, 2 if big else 12)
    reps = 8 if big else 36
    lay = O.IMT_NT if nt else O.IMT_NN
    ven = (lambda A, B, o: torch.matmul(A, B.t(), out=o)) if nt else (lambda A, B, o: torch.matmul(A, B, out=o))
    ours = lambda A, B, o: O.gemm(A, B, lay, out=o)
    tv = min(time_it(ven, s, reps) for _ in range(2))
    to = min(time_it(ours, s, reps) for _ in range(2))
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        ven(*s[0]); torch.cuda.synchronize()
    kn = [e.name for e in prof.events() if e.device_type.name == "CUDA"][:1]
    fl = 2.0 * M * N * K
    print("%-12s %s %5d x %5d x %5d  vendor %7.1f us %4.0f TF | imt_gemm %7.1f us %4.0f TF | %s" %
          (name, "NT" if nt else "NN", M, N, K, tv, fl / tv / 1e6, to, fl / to / 1e6, kn[0][:150] if kn else "?"), flush=True)
    del s
    torch.cuda.empty_cache()
