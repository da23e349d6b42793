#!/usr/bin/env python3
"""Which torch-side ops (not our kernels) run inside a train step, with device time: torch.profiler table."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, build_model, make_batch
from imagetranslate_amd.parallel import train_step
from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup
from torch.profiler import profile, ProfilerActivity
c = CONFIGS["c1"]
model = build_model(c, torch.bfloat16, torch.device("cuda")).train()
opt = AdamInverseSqrtWithWarmup(model.parameters(), lr=1e-4, betas=(0.9, 0.98), warmup_updates=4000)
batch = make_batch(c, 1234, "cuda")
for _ in range(5): train_step(model, opt, batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3): train_step(model, opt, batch)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
