import sys, os
sys.path.insert(0, "/root/repo")
import torch
from imagetranslate_amd import hip_ops as O, _lib as L
lib = L.load()
rows, d = 8192, 512
x = torch.randn(rows, d, device="cuda").bfloat16(); dy = torch.randn(rows, d, device="cuda").bfloat16()
g = torch.randn(d, device="cuda").bfloat16(); b = torch.randn(d, device="cuda").bfloat16()
y, mean, rstd = O.layernorm_fwd(x, g, b)
dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
ws = torch.zeros(32 * 2 * d, device="cuda")
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
lib.imt_debug_spin(256, 256, 0, 3 * 2400 * 1000, st)
for _ in range(64):
    O.layernorm_bwd(dy, x, g, mean, rstd, dg, db, want_dx_drop=True, dx_dropout_p=0.1, dx_dropout_seed=5, partial_ws=ws)
torch.cuda.synchronize()
