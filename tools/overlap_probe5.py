#!/usr/bin/env python3
"""Can weight-gradient GEMMs with a SMALL LDS footprint run under the dX chain?  Stream A: persistent N = 512 GEMMs
(128 KiB LDS, one workgroup per CU).  Stream B: dW products (TN, K = 8192 tokens) on the single-buffer kernel (32 KiB
LDS, 256 threads) or on the grouped kernel's stand-in, the persistent TN kernel (128 KiB).  A alone, B alone, A || B."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O  # noqa: E402

dt = torch.bfloat16
T, d, ff = 8192, 512, 2048
x = torch.randn(T, ff, device="cuda").to(dt); W = (torch.randn(d, ff, device="cuda") * 0.02).to(dt)
outA = torch.empty(T, d, device="cuda", dtype=dt)
dy = torch.randn(T, ff, device="cuda").to(dt); xin = torch.randn(T, d, device="cuda").to(dt)
dW = torch.zeros(ff, d, device="cuda")
NA, NB = 24, 8


def chain_a():
    for _ in range(NA):
        O.gemm(x, W, O.IMT_NT, out=outA)                       # 8192 x 512 x 2048, persistent kernel


def chain_b(variant):
    for _ in range(NB):
        O.gemm(dy, xin, O.IMT_TN, out=dW, accumulate=True, force_general=variant)   # 2048 x 512 x 8192


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both(variant):
    with torch.cuda.stream(sa):
        chain_a()
    with torch.cuda.stream(sb):
        chain_b(variant)


def only(stream, fn):
    with torch.cuda.stream(stream):
        fn()


print("A alone (%d persistent GEMMs)         : %.3f ms" % (NA, timeit(lambda: only(sa, chain_a))))
for variant, name in ((3, "single-buffer 32 KiB"), (1, "double-buffer 64 KiB"), (5, "persistent 128 KiB")):
    b = timeit(lambda: only(sb, lambda: chain_b(variant)))
    ab = timeit(lambda: both(variant))
    print("B alone (%d dW, %-20s): %.3f ms   A || B: %.3f ms" % (NB, name, b, ab))
