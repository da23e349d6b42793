#!/usr/bin/env python3
"""Can weight-gradient work run in the SHADOW of the latency-bound dX chain if its kernel leaves room on the CU?
The grouped launch (512 threads, 128 KiB LDS) cannot share a CU with the chain's persistent kernels (overlap_probe.py:
nothing co-resides).  Here the same four products run on a side stream through small-footprint kernels -- variant 3
(256 threads, 32 KiB LDS, three workgroups per CU) or variant 1 (64 KiB) -- with split-K atomics to make enough workgroups.
Times: chain alone, dW alone, both on two streams (wall clock per repetition, 6 'layers' per repetition)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O

T, d, ff, B, H, S = 8192, 512, 2048, 64, 8, 128
dt = torch.bfloat16
dev = "cuda"
r = lambda *s: torch.randn(*s, device=dev).to(dt)
x, dy, w_o, w_qkv, w1, w2 = r(T, d), r(T, d), r(d, d), r(3 * d, d), r(ff, d), r(d, ff)
dff, z, qkv, dqkv = r(T, ff), r(T, ff), r(T, 3 * d), r(T, 3 * d)
g = torch.ones(d, device=dev).to(dt); bb = torch.zeros(d, device=dev).to(dt)
_, mean, rstd = O.layernorm_fwd(x, g, bb)
dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
o, lse = O.attention_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, 64)
out_d, out_ff, out_dx = torch.empty(T, d, device=dev, dtype=dt), torch.empty(T, ff, device=dev, dtype=dt), torch.empty(T, d, device=dev, dtype=dt)
NL = 6

def chain():
    for _ in range(NL):
        O.layernorm_bwd(dy, x, g, mean, rstd, dg, db)
        O.gemm(dy, w2, O.IMT_NN, out=out_ff, aux=z, aux_mode=O.IMT_AUX_DGELU)
        O.gemm(dff, w1, O.IMT_NN, out=out_d, resid=x)
        O.layernorm_bwd(dy, x, g, mean, rstd, dg, db)
        O.gemm(dy, w_o, O.IMT_NN, out=out_dx)
        O.attention_bwd(dy, qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, B, H, S, S, 64, dq=dqkv[:, :d], dk=dqkv[:, d:2 * d], dv=dqkv[:, 2 * d:])
        O.gemm(dqkv, w_qkv, O.IMT_NN, out=out_dx, resid=x)

gw = [torch.zeros(3 * d, d, device=dev), torch.zeros(d, d, device=dev), torch.zeros(ff, d, device=dev), torch.zeros(d, ff, device=dev)]
gb = [torch.zeros(3 * d, device=dev), torch.zeros(d, device=dev), torch.zeros(ff, device=dev), torch.zeros(d, device=dev)]
prods = [(dqkv, x, 0), (dy, x, 1), (dff, x, 2), (dy, dff, 3)]

def dw_grouped():
    for _ in range(NL):
        O.gemm_grouped_tn([dict(A=a, B=b, out=gw[i], a_colsum=gb[i]) for a, b, i in prods])

def make_dw(variant, split):
    def f():
        for _ in range(NL):
            for a, b, i in prods:
                O.gemm(a, b, O.IMT_TN, out=gw[i], accumulate=True, force_general=variant, split_k=split)
    return f

def wall(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e6

side = torch.cuda.Stream()
def both(dw):
    def f():
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            dw()
            done = torch.cuda.Event(); done.record()
        chain()
        torch.cuda.current_stream().wait_event(done)
    return f

tc = wall(chain)
print("chain alone (%d layers) %.1f us" % (NL, tc), flush=True)
for name, dw in [("grouped (512 thr, 128 KiB)", dw_grouped)] + [("variant %d split-K %d" % (v, s), make_dw(v, s)) for v in (3, 1) for s in (1, 4, 8)]:
    tw = wall(dw)
    tb = wall(both(dw))
    print("%-28s dW alone %7.1f us | sum %7.1f | two streams %7.1f us | hidden %5.1f %% of dW" % (name, tw, tc + tw, tb, 100.0 * (tc + tw - tb) / tw), flush=True)
