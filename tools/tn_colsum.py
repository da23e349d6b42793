import sys, os
sys.path.insert(0, "/root/repo")
import torch
from imagetranslate_amd import hip_ops as O
from tools.gemm_ln_bench import gpu_time
K, M, N = 8128, 30000, 512
A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16()
out = torch.zeros(M, N, device="cuda"); cs = torch.zeros(M, device="cuda")
for colsum in (None, cs):
    ks = gpu_time(lambda: O.gemm(A, B, O.IMT_TN, out=out, accumulate=True, a_colsum=colsum))
    t = sum(ks.values())
    print("vocab dW 30000x512x8128 TN, bias grad %s: %s %.1f us %.0f TFLOP/s" % ("fused" if colsum is not None else "none", "+".join(ks), t, 2.0 * M * N * K / t / 1e6), flush=True)
