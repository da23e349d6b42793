"""Dark time between real kernels: absolute first-workgroup-start / last-workgroup-end stamps (s_memrealtime, one clock for the
whole chip) of 64 consecutive traced launches queued behind a 3-ms spin kernel, so the host is never the pacer.
Run with IMT_TRACE=all IMT_TRACE_RING=1.  Chains: attn_fwd only | persistent GEMM only | attn_fwd, GEMM alternating."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O, _lib as L

lib = L.load()
B, H, T, dh = 64, 8, 128, 64
d = H * dh
M = B * T
qkv = torch.randn(M, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
w = (torch.randn(d, d, device="cuda") / d ** 0.5).bfloat16()
bias = torch.randn(d, device="cuda").bfloat16()
x = torch.randn(M, d, device="cuda").bfloat16()
r = torch.randn(M, d, device="cuda").bfloat16()
out = torch.empty(M, d, device="cuda").bfloat16()
st = torch.cuda.current_stream().cuda_stream


def attn():
    O.attention_fwd(q, k, v, B, H, T, T, dh, dropout_p=0.1, dropout_seed=3)


def gemm():
    O.gemm(x, w, O.IMT_NT, out=out, bias=bias, resid=r, dropout_p=0.1, dropout_seed=5)


mode = sys.argv[1] if len(sys.argv) > 1 else "mixed"
chain = {"attn": [attn] * 64, "gemm": [gemm] * 64, "mixed": [attn, gemm] * 32}[mode]
os.environ.pop("IMT_TRACE_RING_OFF", None)
print("---- chain:", mode, file=sys.stderr, flush=True)
torch.cuda.synchronize()
lib.imt_debug_spin(256, 256, 0, 3 * 2400 * 1000, st)   # ~3 ms at 2.4 GHz: the 64 launches below queue up behind it
for f in chain:
    f()
torch.cuda.synchronize()
