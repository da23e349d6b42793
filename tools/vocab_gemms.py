"""The three vocabulary GEMMs of the C1 step on HBM-cold operands (the logits / d(logits) matrix is 487 MB: it never stays in the
Infinity Cache), per-launch device time; IMT_GEMM_DBG=128 switches the 256-tile kernel's A operand to non-temporal LDS-DMA loads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O, _lib as L
n, V, K = 8128, 30000, 512
dev = "cuda"
x = torch.randn(n, K, device=dev).bfloat16(); w = (torch.randn(V, K, device=dev) * 0.05).bfloat16(); b = torch.zeros(V, device=dev).bfloat16()
dl = [O.alloc_rows(n, V, torch.bfloat16, dev) for _ in range(2)]
for t in dl: t.normal_()
gw = torch.zeros(V, K, device=dev); gb = torch.zeros(V, device=dev)
g = torch.ones(1, device=dev)
lib = L.load()
def prof(fn, reps=6):
    for _ in range(2): fn(0)
    torch.cuda.synchronize(); lib.imt_prof_enable(1)
    for i in range(reps): fn(i)
    torch.cuda.synchronize()
    rows = (L.ProfRow * 64)(); m = lib.imt_prof_report(rows, 64); lib.imt_prof_enable(0)
    return {rows[i].kind.decode(): rows[i].total_ms * 1e3 / reps for i in range(m)}
slabs = torch.empty((4 * n, K), device=dev, dtype=torch.float32)
dx = O.alloc_rows(n, K, torch.bfloat16, dev)
print("env IMT_GEMM_DBG =", os.environ.get("IMT_GEMM_DBG", ""))
print("fwd logits  ", prof(lambda i: O.gemm(x, w, O.IMT_NT, bias=b, out=dl[i % 2])))
print("dX (slabs)  ", prof(lambda i: O.gemm(dl[i % 2], w, O.IMT_NN, out=dx, aux=slabs, aux_mode=O.IMT_AUX_SPLITK_WS, split_k=4, alpha_dev=g)))
print("dW          ", prof(lambda i: O.gemm(dl[i % 2], x, O.IMT_TN, out=gw, accumulate=True, alpha_dev=g, a_colsum=gb)))
