"""Persistent GEMM main loop: full / fill only (IMT_GEMM_DBG=1) / multiply only (=2) / neither (=3), in-kernel phases from
IMT_TRACE=gemm_ws (first tile landed | K loop | epilogue).  Run once per IMT_GEMM_DBG value (the flag is read at first use)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O
for (M, N, K, lay) in [(8192, 512, 2048, O.IMT_NT), (8192, 512, 512, O.IMT_NT), (8192, 512, 2048, O.IMT_NN)]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = (torch.randn(N, K, device="cuda") if lay == O.IMT_NT else torch.randn(K, N, device="cuda")).bfloat16()
    out = torch.empty(M, N, device="cuda").bfloat16()
    for _ in range(3):
        O.gemm(A, B, lay, out=out)
    torch.cuda.synchronize()
