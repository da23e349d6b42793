import os, sys, torch
sys.path.insert(0, "/root/repo")
from imagetranslate_amd import hip_ops as O
def t(lay, M, N, K, fg, bias=True, reps=36):
    g = torch.Generator(device="cuda").manual_seed(1)
    sets = []
    for _ in range(12):
        A = torch.randn((M, K), device="cuda", generator=g).bfloat16()
        B = (torch.randn((N, K) if lay == O.IMT_NT else (K, N), device="cuda", generator=g) * .05).bfloat16()
        sets.append((A, B, torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.randn(N, device="cuda").bfloat16() if bias else None))
    f = lambda A, B, o, b: O.gemm(A, B, lay, out=o, bias=b, force_general=fg)
    for s in sets[:2]: f(*s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps): f(*sets[r % 12])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for lay, M, N, K in [(O.IMT_NT, 8192, 1536, 512), (O.IMT_NT, 8128, 1536, 512), (O.IMT_NT, 8192, 512, 512), (O.IMT_NT, 8192, 512, 2048), (O.IMT_NN, 8192, 512, 1536), (O.IMT_NN, 8192, 512, 2048), (O.IMT_NN, 8192, 512, 512)]:
    print("%s %d x %d x %d: auto %.1f us | v3 %.1f | v5 %.1f | v6 %.1f" % ("NT" if lay == O.IMT_NT else "NN", M, N, K, t(lay, M, N, K, 0), t(lay, M, N, K, 3), t(lay, M, N, K, 5), t(lay, M, N, K, 6)), flush=True)
