"""In-kernel phases of the 256-tile kernel (IMT_TRACE=gemm_xl: first K tile landed | K loop | epilogue) on the step's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O
from tools.gemm_ln_bench import gpu_time
for (M, N, K, lay, aux) in [(8192, 2048, 512, O.IMT_NT, 1), (8192, 2048, 512, O.IMT_NN, 2), (8128, 30000, 512, O.IMT_NT, 0), (8192, 6144, 512, O.IMT_NT, 0)]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = (torch.randn(N, K, device="cuda") if lay == O.IMT_NT else torch.randn(K, N, device="cuda")).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    out = O.alloc_rows(M, N, torch.bfloat16, "cuda")
    z = O.alloc_rows(M, N, torch.bfloat16, "cuda")
    kw = dict(out=out, bias=bias if aux != 2 else None)
    if aux == 1:
        kw.update(aux=z, aux_mode=O.IMT_AUX_GELU_FWD)
    if aux == 2:
        kw.update(aux=z, aux_mode=O.IMT_AUX_DGELU)
    for _ in range(3):
        O.gemm(A, B, lay, **kw)
    torch.cuda.synchronize()
    if not os.environ.get("IMT_TRACE"):
        ks = gpu_time(lambda: O.gemm(A, B, lay, **kw))
        print("%d x %d x %d aux %d: %s" % (M, N, K, aux, ks), flush=True)
