"""One decoder layer's weight-gradient group (7 products, 256 tiles, K = 8192 tokens) and one encoder layer's (4 products,
192 tiles) as single grouped launches; per-launch device time from the library profiler."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O
from tools.gemm_ln_bench import gpu_time
K, d, ff = 8192, 512, 2048


def group(shapes, colsum=True):
    pr = []
    for (m, n) in shapes:
        pr.append(dict(A=torch.randn(K, m, device="cuda").bfloat16(), B=torch.randn(K, n, device="cuda").bfloat16(),
                       out=torch.zeros(m, n, device="cuda"), a_colsum=torch.zeros(m, device="cuda") if colsum else None))
    return pr


for name, shapes in (("decoder layer (256 tiles)", [(3 * d, d), (d, d), (d, d), (2 * d, d), (d, d), (ff, d), (d, ff)]),
                     ("encoder layer (192 tiles)", [(3 * d, d), (d, d), (ff, d), (d, ff)])):
    for colsum, nsets in ((True, 1), (False, 1), (True, 8)):
        sets = [group(shapes, colsum) for _ in range(nsets)]   # 8 sets ~ 0.9 GB of operands: nothing stays in the 256-MB Infinity Cache
        it = [0]

        def run():
            O.gemm_grouped_tn(sets[it[0] % nsets]); it[0] += 1
        ks = gpu_time(run)
        t = sum(ks.values())
        fl = sum(2.0 * K * m * n for m, n in shapes)
        print("%s%s%s: %s %.1f us  %.0f TFLOP/s  (%.3f us per K tile)" % (name, " + bias grads" if colsum else "", " [cold operands]" if nsets > 1 else "", "+".join(ks), t, fl / t / 1e6, t / (K / 64)), flush=True)
