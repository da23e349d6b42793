import sys, os
sys.path.insert(0, "/root/repo")
import torch
from imagetranslate_amd import hip_ops as O
for (M, N, K) in [(8192, 512, 512), (64, 512, 512), (8192, 512, 2048), (8192, 256, 256)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16(); r = torch.randn(M, N, device="cuda").bfloat16()
    for _ in range(3):
        O.gemm_bias_residual_ln(x, w, b, r, b, b)
    torch.cuda.synchronize()
