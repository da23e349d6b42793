"""What a launch costs by geometry: imt_debug_spin (a kernel that does nothing for `cycles` shader clocks) timed with HIP
events over back-to-back launches, for the grid / workgroup / LDS shapes of the step's kernels."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import _lib as L

lib = L.load()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(blocks, threads, lds, cycles, n=200):
    for _ in range(10):
        lib.imt_debug_spin(blocks, threads, lds, cycles, st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        lib.imt_debug_spin(blocks, threads, lds, cycles, st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


for name, blocks, threads, lds in [("256 x 256 thr, 0 LDS", 256, 256, 0), ("256 x 512 thr, 128 KiB (ws GEMM)", 256, 512, 128 << 10),
                                   ("256 x 512 thr, 146 KiB (gemm_ln)", 256, 512, 146 << 10), ("512 x 512 thr, 33 KiB (attention)", 512, 512, 33 << 10),
                                   ("2048 x 256 thr, 0 LDS (LayerNorm)", 2048, 256, 0), ("768 x 256 thr, 32 KiB (sbuf GEMM)", 768, 256, 32 << 10),
                                   ("3776 x 512 thr, 128 KiB (vocabulary projection)", 3776, 512, 128 << 10)]:
    print("%-50s  spin 0: %6.2f us   spin 10 us: %6.2f us" % (name, timed(blocks, threads, lds, 0), timed(blocks, threads, lds, 24000)), flush=True)
