"""Is the step host-bound?  Host time to ENQUEUE a step (no synchronisation) against the GPU time of the step, and a
cProfile of the enqueue work."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import CONFIGS, build_model, make_batch
from imagetranslate_amd.parallel import train_step
from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup

c = CONFIGS["c1"]
dev = torch.device("cuda")
model = build_model(c, torch.bfloat16, dev).train()
opt = AdamInverseSqrtWithWarmup(model.parameters(), lr=1e-4, betas=(0.9, 0.98), warmup_updates=4000)
batches = [make_batch(c, 1234 + 1000 * i, dev, direction=i % 2) for i in range(8)]
k = [0]


def step():
    b = batches[k[0] % 8]; k[0] += 1
    return train_step(model, opt, b, clip=1.0)


for _ in range(8):
    step()
torch.cuda.synchronize()
for trial in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("20 steps: host enqueue %.3f ms/step, wall with GPU %.3f ms/step" % (t_host * 50, t_all * 50), flush=True)
# host cost with the GPU idle in between (pure host path length): sync after each step, time only the enqueue
tot = 0.0
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    tot += time.perf_counter() - t0
print("host enqueue alone (GPU idle at start): %.3f ms/step" % (tot * 50))
import cProfile, pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(28)
