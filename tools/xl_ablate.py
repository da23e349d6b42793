#!/usr/bin/env python3
"""The 256-tile kernel with its in-loop DMA or its MFMAs switched off (force_general = 6 + 100 * dbg; gemm.hip EpiParams.dbg bits 1 / 0):
what each half of the K loop costs alone, HBM-cold operands.  GPU only; results in profiles/r03_gemm_xl_ring.txt."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O
def t(lay, M, N, K, fg, reps=30):
    g = torch.Generator(device="cuda").manual_seed(1)
    sets = []
    for _ in range(10):
        A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
        B = (torch.randn((N, K) if lay == O.IMT_NT else (K, N), device="cuda", generator=g) * .05).bfloat16()
        sets.append((A, B, torch.empty(M, N, device="cuda", dtype=torch.bfloat16)))
    for A, B, o in sets[:2]: O.gemm(A, B, lay, out=o, force_general=fg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        A, B, o = sets[r % 10]; O.gemm(A, B, lay, out=o, force_general=fg)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for lay, M, N, K in [(O.IMT_NT, 8192, 2048, 512), (O.IMT_NT, 8192, 2048, 2048), (O.IMT_NT, 8192, 2048, 8192)]:
    full, nomma, nodma = t(lay, M, N, K, 6), t(lay, M, N, K, 106), t(lay, M, N, K, 206)
    nt = K // 64
    nost, wt = t(lay, M, N, K, 12806), t(lay, M, N, K, 25606)
    print("   no output stores %.1f us | write-through stores %.1f us" % (nost, wt))
    print("NT %d x %d x %d: full %.1f us | DMA only %.1f | MFMA only %.1f   (per 64-deep K tile: %.2f / %.2f / %.2f us incl. fixed parts)" % (M, N, K, full, nomma, nodma, full / nt, nomma / nt, nodma / nt), flush=True)
