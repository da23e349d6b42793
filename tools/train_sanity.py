#!/usr/bin/env python3
"""C1-sized model, bf16, dropout 0.1: memorise 4 fixed synthetic batches for a few hundred steps -- the loss must fall
steadily and stay finite (end-to-end sanity of kernels + optimizer + bf16 shadow at the benchmark size)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, build_model, make_batch
from imagetranslate_amd.parallel import train_step
from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup

c = dict(CONFIGS["c1"])
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
model = build_model(c, torch.bfloat16, torch.device("cuda")).train()
opt = AdamInverseSqrtWithWarmup(model.parameters(), lr=5e-4, betas=(0.9, 0.98), warmup_updates=100)
batches = [make_batch(c, 10 + i, "cuda") for i in range(4)]
t0 = time.time()
hist, mem = [], []
for s in range(steps):
    loss, n = train_step(model, opt, batches[s % 4], clip=1.0)
    if s % 50 == 0 or s == steps - 1:
        hist.append(float(loss.detach()))
        torch.cuda.synchronize()  # nothing in flight: what is still allocated now is retained, not queued
        mem.append(torch.cuda.memory_allocated())
        print("step %4d loss %.4f  held %.3f GB  reserved %.2f GB  (%.1f s)" % (s, hist[-1], mem[-1] / 1e9, torch.cuda.memory_reserved() / 1e9, time.time() - t0), flush=True)
assert all(h == h and h < 1e4 for h in hist), "non-finite loss"
assert hist[-1] < 0.7 * hist[0], "loss did not fall: %s" % hist
assert mem[-1] <= mem[1] * 1.02 + (1 << 20), "memory held between steps grows: %s" % mem
print("ok: %.3f -> %.3f" % (hist[0], hist[-1]))
