#!/usr/bin/env python3
"""Does the cross-attention K|V projection of the encoder states (independent of the decoder's self-attention block)
hide under that block when issued on a second stream?  chain = qkv GEMM -> attention -> out-proj(+dropout+resid) -> LN."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O

T, d, B, H, S = 8192, 512, 64, 8, 128
dt = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.3).to(dt)
x, enc, w_qkv, w_o, w_kv, b = r(T, d), r(T, d), r(3 * d, d), r(d, d), r(2 * d, d), r(d)
g, bb = torch.ones(d, device="cuda").to(dt), torch.zeros(d, device="cuda").to(dt)
qkv, ctx_o, pre, kv = torch.empty(T, 3 * d, device="cuda", dtype=dt), None, torch.empty(T, d, device="cuda", dtype=dt), torch.empty(T, 2 * d, device="cuda", dtype=dt)

def chain():
    O.gemm(x, w_qkv, O.IMT_NT, out=qkv)
    o, _ = O.attention_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, H, S, S, 64, causal=True, dropout_p=0.1, dropout_seed=3)
    O.gemm(o, w_o, O.IMT_NT, out=pre, bias=b, resid=x, dropout_p=0.1, dropout_seed=5)
    O.layernorm_fwd(pre, g, bb)

def side_work():
    O.gemm(enc, w_kv, O.IMT_NT, out=kv)

def wall(fn, reps=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / reps * 1e6

side = torch.cuda.Stream()
def both():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev); side_work()
        done = torch.cuda.Event(); done.record()
    chain()
    torch.cuda.current_stream().wait_event(done)

def serial():
    side_work(); chain()

print("chain %.1f us | kv GEMM %.1f us | serial %.1f us | two streams %.1f us" % (wall(chain), wall(side_work), wall(serial), wall(both)))
