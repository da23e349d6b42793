"""Weight gradients with a SHORT K (few tokens: captioning 992 / 1568 rows): which kernel variant for TN products with many
output tiles?  Per-launch time (events over 30 launches) for variants 1 (register double buffer), 3 (single buffer), 5 (persistent)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K) in [(30000, 512, 992), (512, 2048, 1568), (30000, 512, 320), (2048, 512, 992)]:
    A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16()
    out = torch.zeros(M, N, device="cuda"); cs = torch.zeros(M, device="cuda")
    res = []
    for v in (0, 1, 3, 5):
        try:
            us = t(lambda: O.gemm(A, B, O.IMT_TN, out=out, accumulate=True, a_colsum=cs, force_general=v))
            res.append("v%d %7.1f us %6.0f TF" % (v, us, 2.0 * M * N * K / us / 1e6))
        except Exception as e:
            res.append("v%d failed" % v)
    print("TN %5d x %4d x %4d : %s" % (M, N, K, " | ".join(res)), flush=True)
