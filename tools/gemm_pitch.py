"""Does the row pitch of the operands matter (L2 / HBM channel interleave)?  The same product with operands whose rows are
K elements apart (what the model uses) and K + pad elements apart, per-launch device time from the library's profiler."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O
from tools.gemm_ln_bench import gpu_time


def view(rows, cols, pad):
    full = torch.randn(rows, cols + pad, device="cuda").bfloat16()
    return full[:, :cols]


for (M, N, K, layout) in [(8192, 512, 2048, "NT"), (8192, 512, 512, "NT"), (8192, 2048, 512, "NT"), (8192, 512, 2048, "NN"), (8192, 512, 1536, "NN")]:
    for pa, pb, pc in [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (64, 64, 64), (192, 192, 0), (32, 32, 0)]:
        A = view(M, K, pa)
        if layout == "NT":
            B = view(N, K, pb); lay = O.IMT_NT
        else:
            B = view(K, N, pb); lay = O.IMT_NN
        out = view(M, N, pc)
        ks = gpu_time(lambda: O.gemm(A, B, lay, out=out))
        t = sum(ks.values())
        print("%s %5d x %4d x %4d  pad A %3d B %3d C %3d : %s  %6.1f us  %6.0f TFLOP/s" % (layout, M, N, K, pa, pb, pc, "+".join(ks), t, 2.0 * M * N * K / t / 1e6), flush=True)
