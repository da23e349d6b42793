#!/usr/bin/env python3
"""imt_gemm variants on the decode-step shapes (M = hypotheses = 320): which main loop suits tiny grids."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O
from tools.gemm_epi import dev_time
dt = torch.bfloat16
for (M, N, K, name) in [(320, 1536, 512, "qkv"), (320, 512, 512, "out-proj"), (320, 2048, 512, "ffn1"), (320, 512, 2048, "ffn2"), (320, 30000, 512, "vocab")]:
    A = torch.randn(M, K, device="cuda").to(dt); B = torch.randn(N, K, device="cuda").to(dt)
    bias = torch.randn(N, device="cuda").to(dt); out = torch.empty(M, N, device="cuda", dtype=dt)
    res = []
    for v in (1, 3, 5):
        us = dev_time([lambda: O.gemm(A, B, O.IMT_NT, out=out, bias=bias, force_general=v)], reps=20)
        res.append("%6.1f us" % us)
    print("%-9s M=%d N=%5d K=%4d  dbuf %s | sbuf %s | ws %s" % (name, M, N, K, *res), flush=True)
