#!/usr/bin/env python3
"""Run ONE imt_gemm shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py LAYOUT M N K [split_k] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O  # noqa: E402

lay = {"NT": 0, "NN": 1, "TN": 2}[sys.argv[1]]
M, N, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sk = int(sys.argv[5]) if len(sys.argv) > 5 else 1
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dt = torch.bfloat16
if lay == 0:
    A = torch.randn(M, K, device="cuda").to(dt); B = torch.randn(N, K, device="cuda").to(dt)
elif lay == 1:
    A = torch.randn(M, K, device="cuda").to(dt); B = torch.randn(K, N, device="cuda").to(dt)
else:
    A = torch.randn(K, M, device="cuda").to(dt); B = torch.randn(K, N, device="cuda").to(dt)
out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if lay == 2 else dt)
for _ in range(reps):
    O.gemm(A, B, lay, out=out, split_k=sk, accumulate=(lay == 2 and sk == 1))
torch.cuda.synchronize()
print("done")
