#!/usr/bin/env python3
"""Soak of the one-launch decoder step: many KV-cached beam searches of varying batch / beam / source length on the C1 model; every search
ends with imt_decode_check (BeamDecoder raises if a grid barrier of any step timed out).  GPU only.  python tools/decode_soak.py [searches]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, build_model
from imagetranslate_amd.seq_gen import BeamDecoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = CONFIGS["c1"]
model = build_model(cfg, torch.bfloat16, torch.device("cuda")).eval()
g = torch.Generator().manual_seed(7)
t0, steps = time.time(), 0
for i in range(n):
    B = int(torch.randint(1, 65, (1,), generator=g)); S = int(torch.randint(4, 129, (1,), generator=g)); beam = int(torch.randint(1, 8, (1,), generator=g))
    src = torch.randint(6, cfg["V"], (B, S), generator=g); src[:, 0] = 5; src[:, -1] = 4
    mask = torch.ones(B, S, dtype=torch.bool)
    for b in range(B):
        mask[b, S - (b % 3):] = False
    out = BeamDecoder(model, beam_width=beam, kv_cache=True)(
        src_inputs=src.cuda(), src_sizes=mask.sum(1), first_tokens=torch.full((B,), 5), src_mask=mask.cuda(),
        src_langs=torch.zeros(B, dtype=torch.long).cuda(), tgt_langs=torch.ones(B, dtype=torch.long).cuda(), pad_idx=0, max_len=40)
    steps += max(len(o) for o in out)
    if i % 10 == 9:
        print("%d searches, %d decoding steps, %.1f s" % (i + 1, steps, time.time() - t0), flush=True)
print("ok: %d searches, %d one-launch decoding steps, no abandoned step" % (n, steps))
