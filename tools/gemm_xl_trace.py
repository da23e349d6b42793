#!/usr/bin/env python3
"""Phase time stamps of the 256 x 256-tile GEMM (IMT_TRACE=gemm_xl makes imt_gemm print them to stderr).  Tools only."""
import os
import sys
os.environ["IMT_TRACE"] = "gemm_xl"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O  # noqa: E402
from tools.gemm_xl import operands  # noqa: E402

for lay, M, N, K in [(O.IMT_NT, 8192, 2048, 512), (O.IMT_NT, 8192, 512, 512), (O.IMT_NT, 8192, 512, 2048), (O.IMT_NT, 8192, 2048, 2048),
                     (O.IMT_NT, 256, 256, 512), (O.IMT_NT, 8192, 6144, 512)]:
    A, B = operands(lay, M, N, K, torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        O.gemm(A, B, lay, out=out, force_general=6)
    torch.cuda.synchronize()
