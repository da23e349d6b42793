#!/usr/bin/env python3
"""Ablation of the LDS-DMA GEMM main loop (tuning tool): full / loads only / MFMA only, device time via the library profiler."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O, _lib as L

def dev_us(fn, reps=20):
    lib = L.load()
    for _ in range(3): fn()
    torch.cuda.synchronize(); lib.imt_prof_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    rows = (L.ProfRow * 64)(); n = lib.imt_prof_report(rows, 64); lib.imt_prof_enable(0)
    return sum(rows[i].total_ms for i in range(n)) * 1e3 / reps

dt = torch.bfloat16
for (M, N, K) in [(8192, 512, 2048), (8192, 512, 512), (8192, 2048, 512), (8128, 30000, 512)]:
    A = torch.randn(M, K, device="cuda").to(dt); B = torch.randn(N, K, device="cuda").to(dt)
    out = torch.empty(M, N, device="cuda", dtype=dt)
    res = {}
    for name, code in (("full", 2), ("loads-only", 102), ("mfma-only", 202), ("neither", 302), ("full-4stage", 4), ("loads-only-4stage", 104)):
        res[name] = dev_us(lambda: O.gemm(A, B, O.IMT_NT, out=out, force_general=code))
    print("NT M=%d N=%d K=%d : " % (M, N, K) + "  ".join("%s %.1f us" % kv for kv in res.items()), flush=True)
