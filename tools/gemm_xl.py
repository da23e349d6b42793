#!/usr/bin/env python3
"""256 x 256-tile GEMM (variant 6) against the 128 x 128 kernels on the wide short-K shapes of the C1 step: max error
against torch.matmul (fp32 accumulate) and time per launch.  Tools only."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O  # noqa: E402


def operands(layout, M, N, K, dtype):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K), device="cuda", generator=g).to(dtype)
    B = (torch.randn((N, K) if layout == O.IMT_NT else (K, N), device="cuda", generator=g) * 0.05).to(dtype)
    return A, B


def run(layout, M, N, K, variant, dtype=torch.bfloat16, reps=20, epi=None):
    A, B = operands(layout, M, N, K, dtype)
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    kw = {}
    if epi == "gelu":
        kw = dict(bias=torch.randn(N, device="cuda").to(dtype), aux=torch.empty(M, N, device="cuda", dtype=dtype), aux_mode=O.IMT_AUX_GELU_FWD)
    elif epi == "dgelu":
        kw = dict(aux=torch.randn(M, N, device="cuda").to(dtype), aux_mode=O.IMT_AUX_DGELU)
    elif epi == "resid":
        kw = dict(bias=torch.randn(N, device="cuda").to(dtype), resid=torch.randn(M, N, device="cuda").to(dtype), dropout_p=0.1, dropout_seed=7)
    for _ in range(3):
        O.gemm(A, B, layout, out=out, force_general=variant, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        O.gemm(A, B, layout, out=out, force_general=variant, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    err = None
    if epi is None:
        ref = A.float() @ (B.float().t() if layout == O.IMT_NT else B.float())
        err = float((out.float() - ref).abs().max() / ref.abs().max())
    return us, 2.0 * M * N * K / us / 1e6, err


def main():
    names = {0: "NT", 1: "NN"}
    shapes = [(O.IMT_NT, 8192, 2048, 512), (O.IMT_NN, 8192, 2048, 512), (O.IMT_NT, 8128, 2048, 512), (O.IMT_NT, 8192, 1536, 512),
              (O.IMT_NN, 8192, 1536, 512), (O.IMT_NT, 8192, 6144, 512), (O.IMT_NT, 8128, 30000, 512), (O.IMT_NT, 8192, 512, 2048),
              (O.IMT_NT, 8192, 512, 512), (O.IMT_NT, 1000, 700, 192)]
    for lay, M, N, K in shapes:
        line = "%s %5d x %5d x %5d " % (names[lay], M, N, K)
        for v in (3, 6, 6406):
            us, tf, err = run(lay, M, N, K, v)
            line += "| v%d %6.1f us %4.0f TF %.0e " % (v, us, tf, err)
        print(line, flush=True)
    for epi in ("gelu", "dgelu", "resid"):
        for lay in (O.IMT_NT, O.IMT_NN):
            line = "%s 8192 x 2048 x 512 %-5s " % (names[lay], epi)
            for v in (3, 6, 6406):
                us, tf, _ = run(lay, 8192, 2048, 512, v, epi=epi)
                line += "| v%d %7.1f us %5.0f TF " % (v, us, tf)
            print(line, flush=True)
    # fp32 cross-check of the tile arithmetic
    for lay in (O.IMT_NT, O.IMT_NN):
        us, tf, err = run(lay, 1000, 776, 96, 6, dtype=torch.float32)
        print("fp32 %s 1000 x 776 x 96 v6 err %.2e" % (names[lay], err))


if __name__ == "__main__":
    main()
