#!/usr/bin/env python3
"""How does the C1 train step behave when some CUs are held by a long-running kernel on another stream (as RCCL's
all-reduce kernels do during the backward of a data-parallel job)?  A spin kernel occupies K workgroups on a side
stream for the whole step; step time is compared with the undisturbed one."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, build_model, make_batch
from imagetranslate_amd import _lib as L
from imagetranslate_amd.parallel import train_step
from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup

c = CONFIGS["c1"]
model = build_model(c, torch.bfloat16, torch.device("cuda")).train()
opt = AdamInverseSqrtWithWarmup(model.parameters(), lr=1e-4, betas=(0.9, 0.98), warmup_updates=4000)
batch = make_batch(c, 1234, "cuda")
lib = L.load()
side = torch.cuda.Stream()

def run(blocks, threads=256, lds=4096, steps=20):
    for _ in range(5): train_step(model, opt, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if blocks:
            side.wait_stream(torch.cuda.current_stream())
            # ~6 ms of spinning per step at ~2.1 GHz: covers most of the step
            L.check(lib.imt_debug_spin(blocks, threads, lds, int(6e-3 * 2.1e9), ctypes.c_void_p(side.cuda_stream)), "spin")
        train_step(model, opt, batch)
        if blocks:
            torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

base = run(0)
print("undisturbed: %.3f ms/step" % base, flush=True)
for blocks, threads, lds in ((8, 256, 4096), (32, 256, 4096), (64, 256, 4096), (32, 512, 32768), (64, 512, 32768), (32, 512, 49152)):
    ms = run(blocks, threads, lds)
    print("%3d workgroups held on a side stream (%d threads, %d KiB LDS): %.3f ms/step (%+.1f%%)" %
          (blocks, threads, lds // 1024, ms, 100 * (ms / base - 1)), flush=True)

if os.environ.get("IMT_COMM_KINDS") != "1":
    raise SystemExit(0)
# per-kernel view: which kinds lose time with 32 workgroups held
def kinds(blocks):
    for _ in range(3): train_step(model, opt, batch)
    torch.cuda.synchronize()
    lib.imt_prof_enable(1)
    for _ in range(3):
        if blocks:
            side.wait_stream(torch.cuda.current_stream())
            L.check(lib.imt_debug_spin(blocks, 256, 4096, int(6e-3 * 2.1e9), ctypes.c_void_p(side.cuda_stream)), "spin")
        train_step(model, opt, batch)
        if blocks:
            torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    rows = (L.ProfRow * 256)()
    n = lib.imt_prof_report(rows, 256)
    lib.imt_prof_enable(0)
    return {rows[i].kind.decode(): rows[i].total_ms / 3 for i in range(n)}

a, b = kinds(0), kinds(32)
print("%-28s %9s %9s" % ("kind", "free", "32 held"))
for k in sorted(a, key=lambda k: -(b.get(k, 0) - a[k])):
    print("%-28s %9.3f %9.3f  %+.3f" % (k, a[k], b.get(k, 0.0), b.get(k, 0.0) - a[k]))
