"""Driver for tools/pmc_attn.sh: a few launches of the C1 attention kernels (and the LayerNorm backward) with cold-ish operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetranslate_amd import hip_ops as O
B, H, T, dh = 64, 8, 128, 64
d = H * dh
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
causal = int(os.environ.get("CAUSAL", "0"))
for p in (0.1,):
    for _ in range(4):
        o, lse = O.attention_fwd(q, k, v, B, H, T, T, dh, dropout_p=p, dropout_seed=3, causal=causal)
    do = torch.randn_like(o)
    for _ in range(4):
        O.attention_bwd(do, q, k, v, o, lse, B, H, T, T, dh, dropout_p=p, dropout_seed=3, causal=causal)
x = torch.randn(B * T, d, device="cuda").bfloat16()
g = torch.ones(d, device="cuda").bfloat16(); bb = torch.zeros(d, device="cuda").bfloat16()
y, mean, rstd = O.layernorm_fwd(x, g, bb)
dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
for _ in range(4):
    O.layernorm_bwd(x, x, g, mean, rstd, dg, db)
torch.cuda.synchronize()
