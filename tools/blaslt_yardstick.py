#!/usr/bin/env python3
"""Yardstick only (NOT used by the product): what the vendor library (torch.matmul -> hipBLASLt) reaches on the C1
GEMM shapes, to size the headroom of the hand-written kernels."""
import torch

def bench(M, N, K, nt=True, reps=30):
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    B = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) if nt else torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    f = (lambda: A @ B.t()) if nt else (lambda: A @ B)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K / us / 1e6

T, d, ff, V = 8192, 512, 2048, 30000
for name, M, N, K, nt in [("qkv fwd", T, 3 * d, d, True), ("attn-out fwd", T, d, d, True), ("ffn1 fwd", T, ff, d, True),
                          ("ffn2 fwd", T, d, ff, True), ("vocab fwd", 8128, V, d, True), ("qkv dx", T, d, 3 * d, False),
                          ("ffn1 dx", T, d, ff, False), ("ffn2 dx", T, ff, d, False), ("vocab dx", 8128, d, V, False)]:
    us, tf = bench(M, N, K, nt)
    print("%-14s %s M=%5d N=%5d K=%5d  hipBLASLt %7.1f us %5.0f TF/s" % (name, "NT" if nt else "NN", M, N, K, us, tf), flush=True)
A = torch.randn(T, 3 * d, device="cuda", dtype=torch.bfloat16); X = torch.randn(T, d, device="cuda", dtype=torch.bfloat16)
for _ in range(5): A.t() @ X
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): A.t() @ X
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 30
print("qkv dW         TN M= 1536 N=  512 K= 8192  hipBLASLt %7.1f us %5.0f TF/s" % (us, 2.0 * 1536 * 512 * 8192 / us / 1e6))
