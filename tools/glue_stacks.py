#!/usr/bin/env python3
"""Where do the torch-side device ops of a train step (copies, fills, small elementwise kernels) come from?  torch.profiler with
stacks over 2 steps of the bench loop (8 rotating batches as bench.py); prints every non-library device op with its python stack."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, build_model, make_batch
from imagetranslate_amd.parallel import train_step
from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup
from torch.profiler import profile, ProfilerActivity
c = CONFIGS["c1"]
model = build_model(c, torch.bfloat16, torch.device("cuda")).train()
opt = AdamInverseSqrtWithWarmup(model.parameters(), lr=1e-4, betas=(0.9, 0.98), warmup_updates=4000)
batches = [make_batch(c, 1234 + 1000 * i, "cuda", direction=i % 2) for i in range(4)]
k = [0]
def step():
    b = batches[k[0] % 4]; k[0] += 1
    train_step(model, opt, b, active_head=int(b["dst_langs"][0]))
for _ in range(6): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2): step()
    torch.cuda.synchronize()
evs = prof.events()
seen = {}
for e in evs:
    if e.device_type.name != "CPU":
        continue
    name = e.name
    if not (name.startswith("aten::") or "Memcpy" in name or "Memset" in name or "hipMem" in name):
        continue
    dev_us = e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
    if dev_us <= 0 or any(ch.name.startswith("aten::") for ch in e.cpu_children if (ch.device_time_total if hasattr(ch, "device_time_total") else ch.cuda_time_total) > 0):
        continue  # report leaves only
    st = [s for s in (e.stack or []) if "imagetranslate_amd" in s or "bench.py" in s or "tools/" in s][:3]
    key = (name, tuple(st))
    a = seen.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += dev_us
for (name, st), (n, us) in sorted(seen.items(), key=lambda kv: -kv[1][1]):
    print("%-28s x%-3d %7.1f us/step  %s" % (name, n // 2, us / 2, " <- ".join(s.split("/")[-1] for s in st)))
