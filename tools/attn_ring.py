import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O, _lib as L
lib = L.load()
B, H, T, dh = 64, 8, 128, 64
d = H * dh
qkv = torch.randn(B * T, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
o, lse = O.attention_fwd(q, k, v, B, H, T, T, dh, dropout_p=0.1, dropout_seed=3)
do = torch.randn_like(o)
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
lib.imt_debug_spin(256, 256, 0, 3 * 2400 * 1000, st)
for _ in range(64):
    O.attention_bwd(do, q, k, v, o, lse, B, H, T, T, dh, dropout_p=0.1, dropout_seed=3)
torch.cuda.synchronize()
