"""Attention forward / backward at the captioning shapes (32 images, 31-token captions, 49 regions): per-launch event times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O
from imagetranslate_amd import _lib as L
lib = L.load()


def gpu_time(fn, n=20):
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
    return {rows[i].kind.decode(): round(rows[i].total_ms * 1e3 / max(1, rows[i].launches), 1) for i in range(k)}


H, dh = 8, 64
d = H * dh
for (B, Tq, Tk, causal) in [(32, 31, 31, True), (32, 31, 49, False), (64, 127, 127, True), (64, 127, 128, False), (64, 64, 64, False), (64, 65, 65, False)]:
    q = torch.randn(B * Tq, d, device="cuda").bfloat16()
    k = torch.randn(B * Tk, d, device="cuda").bfloat16()
    v = torch.randn(B * Tk, d, device="cuda").bfloat16()
    qm = torch.ones(B, Tq, dtype=torch.uint8, device="cuda") if causal else None
    for p in (0.0, 0.1):
        o, lse = O.attention_fwd(q, k, v, B, H, Tq, Tk, dh, query_mask=qm, causal=causal, dropout_p=p, dropout_seed=3)
        do = torch.randn_like(o)
        tf = gpu_time(lambda: O.attention_fwd(q, k, v, B, H, Tq, Tk, dh, query_mask=qm, causal=causal, dropout_p=p, dropout_seed=3))
        tb = gpu_time(lambda: O.attention_bwd(do, q, k, v, o, lse, B, H, Tq, Tk, dh, query_mask=qm, causal=causal, dropout_p=p, dropout_seed=3))
        print("B %3d Tq %3d Tk %3d causal %d p %.1f | fwd %s | bwd %s" % (B, Tq, Tk, causal, p, tf, tb), flush=True)
