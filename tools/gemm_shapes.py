#!/usr/bin/env python3
"""Micro-benchmark of imt_gemm on every GEMM shape of the C1 train step (tools only; not part of the product).
Prints per-shape time and TFLOP/s so that tile / split-K choices are made from measurements."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O  # noqa: E402


def bench(layout, M, N, K, dtype=torch.bfloat16, split_k=1, reps=20, **kw):
    dev = "cuda"
    if layout == O.IMT_NT:
        A = torch.randn(M, K, device=dev).to(dtype); B = torch.randn(N, K, device=dev).to(dtype)
    elif layout == O.IMT_NN:
        A = torch.randn(M, K, device=dev).to(dtype); B = torch.randn(K, N, device=dev).to(dtype)
    else:
        A = torch.randn(K, M, device=dev).to(dtype); B = torch.randn(K, N, device=dev).to(dtype)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if layout == O.IMT_TN else dtype)
    for _ in range(3):
        O.gemm(A, B, layout, out=out, split_k=split_k, accumulate=(layout == O.IMT_TN and split_k == 1), **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        O.gemm(A, B, layout, out=out, split_k=split_k, accumulate=(layout == O.IMT_TN and split_k == 1), **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, 2.0 * M * N * K / us / 1e6


def main():
    T, d, ff, V = 8192, 512, 2048, 30000
    names = {0: "NT", 1: "NN", 2: "TN"}
    shapes = [
        (O.IMT_NT, T, 3 * d, d, "qkv fwd"), (O.IMT_NT, T, d, d, "attn-out fwd"), (O.IMT_NT, T, ff, d, "ffn1 fwd"),
        (O.IMT_NT, T, d, ff, "ffn2 fwd"), (O.IMT_NT, 8128, V, d, "vocab fwd"),
        (O.IMT_NN, T, d, 3 * d, "qkv dx"), (O.IMT_NN, T, d, d, "attn-out dx"), (O.IMT_NN, T, d, ff, "ffn1 dx"),
        (O.IMT_NN, T, ff, d, "ffn2 dx"), (O.IMT_NN, 8128, d, V, "vocab dx"),
    ]
    for lay, M, N, K, name in shapes:
        res = []
        for variant in (1, 2, 3, 5):
            if variant in (2, 5) and K % 64:
                res.append((0.0, 0.0)); continue
            res.append(bench(lay, M, N, K, force_general=variant))
        print("%-14s %s M=%5d N=%5d K=%5d  dbuf %6.1f us %5.0f TF | dma3 %6.1f us %5.0f TF | sbuf %6.1f us %5.0f TF | ws %6.1f us %5.0f TF" % (
            name, names[lay], M, N, K, res[0][0], res[0][1], res[1][0], res[1][1], res[2][0], res[2][1], res[3][0], res[3][1]), flush=True)
    for (M, N, name) in [(3 * d, d, "qkv dW"), (d, d, "attn-out dW"), (ff, d, "ffn1 dW"), (d, ff, "ffn2 dW")]:
        for sk in (1, 4):
            us, tf = bench(O.IMT_TN, M, N, T, split_k=sk)
            print("%-14s TN M=%5d N=%5d K=%5d split_k=%2d %8.1f us  %7.1f TF/s" % (name, M, N, T, sk, us, tf), flush=True)
    for sk in (1, 2, 4):
        us, tf = bench(O.IMT_TN, V, d, 8128, split_k=sk)
        print("%-14s TN M=%5d N=%5d K=%5d split_k=%2d %8.1f us  %7.1f TF/s" % ("vocab dW", V, d, 8128, sk, us, tf), flush=True)


if __name__ == "__main__":
    main()
