"""LayerNorm backward as the stack runs it (partial-sum workspace, dropout twin output) on operands that do NOT stay in the
Infinity Cache (24 rotating (x, dy) pairs = 400 MB): per-launch device time from the library's profiler.  A/B over the
environment: IMT_LN_RB4=1 restores the round-2 four-row batches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O, _lib as L
rows, d, NP = 8192, 512, 24
dt = torch.bfloat16
xs = [torch.randn(rows, d, device="cuda").to(dt) for _ in range(NP)]
dys = [torch.randn(rows, d, device="cuda").to(dt) for _ in range(NP)]
g = torch.ones(d, device="cuda").to(dt); b = torch.zeros(d, device="cuda").to(dt)
_, mean, rstd = O.layernorm_fwd(xs[0], g, b)
dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
ws = torch.zeros(O.LN_PARTIAL_COPIES * 2 * d, device="cuda")
lib = L.load()
def run(n, **kw):
    for i in range(n):
        O.layernorm_bwd(dys[i % NP], xs[i % NP], g, mean, rstd, dg, db, partial_ws=ws, **kw)
for name, kw in (("plain", {}), ("+dropout twin", dict(want_dx_drop=True, dx_dropout_p=0.1, dx_dropout_seed=5))):
    run(NP, **kw); torch.cuda.synchronize()
    lib.imt_prof_enable(1); run(4 * NP, **kw); torch.cuda.synchronize()
    rws = (L.ProfRow * 64)(); n = lib.imt_prof_report(rws, 64); lib.imt_prof_enable(0)
    t = sum(rws[i].total_ms for i in range(n)) * 1e3 / (4 * NP)
    nbytes = (4 if kw else 3) * rows * d * 2
    print("ln_bwd %-14s %6.2f us/launch  %6.0f GB/s  (IMT_LN_RB4=%s)" % (name, t, nbytes / t / 1e3, os.environ.get("IMT_LN_RB4", "")), flush=True)
