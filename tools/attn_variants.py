"""Attention fwd / fused bwd at C1 shapes: time per launch (HIP events over 50 back-to-back launches) for dropout on/off,
causal on/off, key mask on/off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O
B, H, T, dh = 64, 8, 128, 64
d = H * dh
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
km = torch.ones(B, T, dtype=torch.uint8, device="cuda")

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for p in (0.0, 0.1):
    for causal in (0, 1):
        for mask in (None, km):
            o, lse = O.attention_fwd(q, k, v, B, H, T, T, dh, dropout_p=p, dropout_seed=3, causal=causal, key_mask=mask)
            do = torch.randn_like(o)
            tf = t(lambda: O.attention_fwd(q, k, v, B, H, T, T, dh, dropout_p=p, dropout_seed=3, causal=causal, key_mask=mask))
            tb = t(lambda: O.attention_bwd(do, q, k, v, o, lse, B, H, T, T, dh, dropout_p=p, dropout_seed=3, causal=causal, key_mask=mask))
            print("dropout %.1f causal %d key_mask %d : fwd %6.2f us  bwd %6.2f us" % (p, causal, mask is not None, tf, tb), flush=True)
