#!/usr/bin/env python3
"""Does a layer's grouped weight-gradient GEMM overlap with the latency-bound dX chain of the next layer when they
run on two HIP streams?  Times: chain alone, dW alone, both concurrently (wall clock over many repetitions)."""
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

def chain():  # one decoder-ish layer of dX work: LN bwd, ff2 dx (GELU'), ff1 dx, LN bwd, o dx, attention bwd, qkv dx
    O.layernorm_bwd(dy, x, g, mean, rstd, dg, db)
    O.gemm(dy, w2, O.IMT_NN, out=out_ff, aux=z, aux_mode=O.IMT_AUX_DGELU)
    O.gemm(dff, w1, O.IMT_NN, out=out_d, resid=x)
    O.layernorm_bwd(dy, x, g, mean, rstd, dg, db)
    O.gemm(dy, w_o, O.IMT_NN, out=out_dx)
    O.attention_bwd(dy, qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, B, H, S, S, 64, dq=dqkv[:, :d], dk=dqkv[:, d:2 * d], dv=dqkv[:, 2 * d:])
    O.gemm(dqkv, w_qkv, O.IMT_NN, out=out_dx, resid=x)

gw = [torch.zeros(3 * d, d, device=dev), torch.zeros(d, d, device=dev), torch.zeros(ff, d, device=dev), torch.zeros(d, ff, device=dev)]
gb = [torch.zeros(3 * d, device=dev), torch.zeros(d, device=dev), torch.zeros(ff, device=dev), torch.zeros(d, device=dev)]
def dw():
    O.gemm_grouped_tn([dict(A=dqkv, B=x, out=gw[0], a_colsum=gb[0]), dict(A=dy, B=x, out=gw[1], a_colsum=gb[1]),
                       dict(A=dff, B=x, out=gw[2], a_colsum=gb[2]), dict(A=dy, B=dff, out=gw[3], a_colsum=gb[3])])

def wall(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e6

side = torch.cuda.Stream()
def both():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        dw()
        done = torch.cuda.Event(); done.record()
    chain()
    torch.cuda.current_stream().wait_event(done)

tc, tw = wall(chain), wall(dw)
tb = wall(both)
print("chain alone %.1f us | dW alone %.1f us | sum %.1f us | two streams %.1f us" % (tc, tw, tc + tw, tb))
