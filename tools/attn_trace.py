import sys
sys.path.insert(0, "/root/repo")
import torch
from imagetranslate_amd import hip_ops as O
B, H, T, dh = 64, 8, 128, 64
d = H * dh
qkv = torch.randn(B * T, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
for p in (0.0, 0.1):
    for _ in range(3):
        o, lse = O.attention_fwd(q, k, v, B, H, T, T, dh, dropout_p=p, dropout_seed=3)
    do = torch.randn_like(o)
    for _ in range(3):
        O.attention_bwd(do, q, k, v, o, lse, B, H, T, T, dh, dropout_p=p, dropout_seed=3)
    torch.cuda.synchronize()
    print("---- p", p, file=sys.stderr)
