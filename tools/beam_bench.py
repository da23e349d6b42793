"""Beam-search throughput on the C1 model shape (6+6 layers, d=512, V=30000): KV-cached incremental decoding vs
the reference's per-step recomputation, both on the HIP path.  Usage: python tools/beam_bench.py [B] [S] [beam]"""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, build_model  # noqa: E402
from imagetranslate_amd.seq_gen import BeamDecoder  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    beam = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cfg = CONFIGS[os.environ.get("BEAM_BENCH_CONFIG", "c1")]   # e.g. ref768: the reference's default --embed 768 / 12 heads / 3072
    model = build_model(cfg, torch.bfloat16, torch.device("cuda")).eval()
    g = torch.Generator().manual_seed(1234)
    src = torch.randint(6, cfg["V"], (B, S), generator=g)
    src[:, 0] = 5
    src[:, -1] = 4
    mask = torch.ones(B, S, dtype=torch.bool)
    args = dict(src_inputs=src.cuda(), src_sizes=torch.full((B,), S), first_tokens=torch.full((B,), 5), src_mask=mask.cuda(),
                src_langs=torch.zeros(B, dtype=torch.long).cuda(), tgt_langs=torch.ones(B, dtype=torch.long).cuda(), pad_idx=0)
    from imagetranslate_amd import _lib as L
    lib = L.load()
    dec = BeamDecoder(model, beam_width=beam, kv_cache=True)
    dec(max_len=8, **args)
    torch.cuda.synchronize()
    lib.imt_prof_enable(1)
    out = dec(**args)
    torch.cuda.synchronize()
    rows = (L.ProfRow * 64)()
    n = lib.imt_prof_report(rows, 64)
    lib.imt_prof_enable(0)
    steps = max(len(o) for o in out)
    print("per-kernel device time of the cached search (%d steps):" % steps)
    for r in sorted(rows[:n], key=lambda r: -r.total_ms):
        print("  %-28s %6d launches %8.3f ms  (%.1f us each)" % (r.kind.decode(), r.launches, r.total_ms, 1e3 * r.total_ms / r.launches))
    print("  total %.3f ms" % sum(r.total_ms for r in rows[:n]), flush=True)
    runs = ((True, None, "1"), (True, None, "0"), (False, None, "0"))
    if os.environ.get("BEAM_BENCH_CACHED_ONLY"):   # counter passes (tools/collect_traffic_decode.sh): skip the 15-s recomputing search
        runs = runs[:2]
    for kv, max_len, fused in runs:
        if kv:
            print("IMT_DECODE_FUSED=%s (1: one launch per decoder step, 0: the launch-per-operator chain)" % fused)
        os.environ["IMT_DECODE_FUSED"] = fused
        dec = BeamDecoder(model, beam_width=beam, kv_cache=kv)
        out = dec(max_len=8, **args)  # warm-up
        torch.cuda.synchronize()
        t0 = time.time()
        out = dec(max_len=max_len, **args)
        torch.cuda.synchronize()
        dt = time.time() - t0
        n = sum(len(o) for o in out)
        print("kv_cache=%s B=%d S=%d beam=%d: %.3f s, %d output tokens, %.0f tok/s, %.2f ms/step" %
              (kv, B, S, beam, dt, n, n / dt, 1e3 * dt / max(len(o) for o in out)), flush=True)


if __name__ == "__main__":
    main()
