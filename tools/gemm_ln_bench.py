"""Fused dense + bias + dropout + residual + LayerNorm (imt_gemm_bias_residual_ln, 32-row row-complete tiles) against the two
launches it replaces (imt_gemm with the residual epilogue, then imt_layernorm_fwd), per shape; HIP events over 50
back-to-back pairs / launches (each figure includes the launch boundaries of its own launches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O


def gpu_time(fn, n=30):
    """per-kind GPU time per call of fn from the library's per-launch event profiler (what bench.py's roofline uses)"""
    from imagetranslate_amd import _lib as L
    lib = L.load()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.imt_prof_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    rows = (L.ProfRow * 64)()
    k = lib.imt_prof_report(rows, 64)
    lib.imt_prof_enable(0)
    return {rows[i].kind.decode(): rows[i].total_ms * 1e3 / max(1, rows[i].launches) for i in range(k)}


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    dev = "cuda"
    for dtype in (torch.bfloat16,):
        for (M, N, K) in [(8192, 512, 512), (8128, 512, 512), (8192, 512, 2048), (8128, 512, 2048), (320, 512, 512), (320, 512, 2048),
                          (64, 512, 512), (2048, 512, 512), (8192, 256, 256), (8192, 128, 128)]:
            x = torch.randn(M, K, device=dev).to(dtype)
            w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dtype)
            b = torch.randn(N, device=dev).to(dtype)
            r = torch.randn(M, N, device=dev).to(dtype)
            g = torch.randn(N, device=dev).to(dtype)
            be = torch.randn(N, device=dev).to(dtype)
            for p in (0.0, 0.1):
                sep = timed(lambda: O.layernorm_fwd(O.gemm(x, w, O.IMT_NT, bias=b, resid=r, dropout_p=p, dropout_seed=5), g, be))
                fus = timed(lambda: O.gemm_bias_residual_ln(x, w, b, r, g, be, dropout_p=p, dropout_seed=5))
                ks = gpu_time(lambda: O.layernorm_fwd(O.gemm(x, w, O.IMT_NT, bias=b, resid=r, dropout_p=p, dropout_seed=5), g, be))
                kf = gpu_time(lambda: O.gemm_bias_residual_ln(x, w, b, r, g, be, dropout_p=p, dropout_seed=5))
                t_sep = sum(ks.values())
                t_fus = sum(kf.values())
                print("%s %5d x %4d x %4d p=%.1f | host-paced: gemm+LN %6.1f us, fused %6.1f us | per-launch events: %s = %5.1f us ; fused %5.1f us  %5.0f TFLOP/s | fused/separate %.2f"
                      % (str(dtype)[6:], M, N, K, p, sep, fus, " + ".join("%s %.1f" % kv for kv in ks.items()), t_sep, t_fus,
                         2.0 * M * N * K / t_fus / 1e6, t_fus / t_sep), flush=True)


if __name__ == "__main__":
    main()
