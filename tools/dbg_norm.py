import sys, os
sys.path.insert(0, "/root/repo")
import torch
from tests.test_gpu_model import _pair, _toy_batch
from imagetranslate_amd.parallel import train_step
from imagetranslate_amd.utils import build_optimizer
from imagetranslate_amd import hip_ops as O
_, ours = _pair(seed=3)
ours.set_compute_dtype(torch.bfloat16); ours.eval()
opt = build_optimizer(ours, 2e-3, 2)
from imagetranslate_amd.param_store import store_of
st = store_of(ours.encoder).ensure()
for s in (11, 12, 13):
    b = _toy_batch(seed=s)
    b = {k: v.cuda() if k not in ("src_langs", "dst_langs") else v for k, v in b.items()}
    loss, _ = ours.loss_fused(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
    loss.backward()
    torch.cuda.synchronize()
    full = float((st.grad.double() ** 2).sum())
    part = opt._partial
    segs = opt._segments(st)
    print("step", s, "segments", segs, "partial", None if part is None else part[0], "generation", st.grad_generation)
    n = opt._grad_norm_sq(st)
    torch.cuda.synchronize()
    print("   full (fp64 torch) %.9e   scheme %.9e" % (full, float(n)))
    opt.step(max_grad_norm=0.5, zero_grad=True)
    torch.cuda.synchronize()
