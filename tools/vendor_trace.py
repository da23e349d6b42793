#!/usr/bin/env python3
"""Yardstick only: device duration of the vendor kernel (torch.matmul -> hipBLASLt) and of imt_gemm's on the FFN-up shape with HBM-cold
operands, from torch.profiler's kernel records (kernel start to kernel end, without the gap between launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O
from torch.profiler import profile, ProfilerActivity
for M, N, K in [(8192, 2048, 512), (8192, 2048, 2048)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    s = []
    for _ in range(12):
        A = torch.randn(M, K, device="cuda", generator=g).bfloat16(); B = (torch.randn(N, K, device="cuda", generator=g) * .05).bfloat16()
        s.append((A, B, torch.empty(M, N, device="cuda", dtype=torch.bfloat16)))
    for r in range(12):
        A, B, o = s[r]; torch.matmul(A, B.t(), out=o); O.gemm(A, B, O.IMT_NT, out=o)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for r in range(36):
            A, B, o = s[r % 12]; torch.matmul(A, B.t(), out=o)
        torch.cuda.synchronize()
        for r in range(36):
            A, B, o = s[r % 12]; O.gemm(A, B, O.IMT_NT, out=o)
        torch.cuda.synchronize()
    print("NT %d x %d x %d" % (M, N, K))
    for e in prof.key_averages():
        if e.device_type.name == "CUDA" and e.count >= 30:
            t = e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
            print("  %6.1f us x %3d  %s" % (t / e.count, e.count, e.key[:110]), flush=True)
