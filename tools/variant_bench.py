"""One product on each kernel variant (force_general: 3 single-buffer, 5 persistent wave-specialised, 6 256-tile), per-launch
device time from the library profiler.  usage: variant_bench.py M N K [NT|NN]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O
from tools.gemm_ln_bench import gpu_time
shapes = [(8192, 1536, 512, "NT"), (8192, 2048, 512, "NT"), (8192, 512, 512, "NT"), (8192, 512, 1536, "NN"), (8128, 30000, 512, "NT")]
if len(sys.argv) >= 4:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "NT")]
for (M, N, K, lay) in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = (torch.randn(N, K, device="cuda") if lay == "NT" else torch.randn(K, N, device="cuda")).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    out = O.alloc_rows(M, N, torch.bfloat16, "cuda")
    L = O.IMT_NT if lay == "NT" else O.IMT_NN
    res = []
    for v in (0, 3, 5, 6):
        try:
            ks = gpu_time(lambda: O.gemm(A, B, L, out=out, bias=bias, force_general=v))
            t = sum(ks.values())
            res.append("%d:%s %.1f us %.0f TF" % (v, "+".join(k.replace("gemm_", "").replace("_bf16", "") for k in ks), t, 2.0 * M * N * K / t / 1e6))
        except Exception as e:
            res.append("%d: n/a" % v)
    print("%s %d x %d x %d | " % (lay, M, N, K) + " | ".join(res), flush=True)
