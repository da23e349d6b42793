#!/usr/bin/env python3
"""The two secondary workloads of SURVEY section 8(d) on one MI355X (the headline C1 stays in bench.py):
  C3 captioning: region features [32,49,2048] -> fc -> 6-layer decoder d=512 -> captions [32,32]   (0.265 TFLOP/step)
  C4 MASS      : monolingual src [64,256], span of int(255/2) tokens masked on the device, 6L/6L d=512 (4.27 TFLOP/step)
Prints one JSON line per workload: ms/step, target tokens/s, TFLOP/s against the algorithmic FLOP counts of the survey."""
import json, os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd.image_model import ImageCaptioning, ImageMassSeq2Seq
from imagetranslate_amd.textprocessor import SyntheticTextProcessor
from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup, mass_mask_device

V, d, ff, heads = 30000, 512, 2048, 8
tp = SyntheticTextProcessor(V)
dev = torch.device("cuda")


def timed(step, warmup=5, steps=20):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        n = step()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    if "--breakdown" in sys.argv:  # per-kernel-kind table of the library's launch timer, to stderr
        from bench import profile_pass
        for r in profile_pass(step)[:14]:
            print("    %-28s %4d launches %8.3f ms/step %7.1f TFLOP/s" % (r["kind"], r["launches"], r["ms"], r["flops"] / max(r["ms"], 1e-9) / 1e9), file=sys.stderr)
    return sec, n


def c3():
    torch.manual_seed(1234)
    m = ImageCaptioning(tp, lang_dec=False, enc_layer=6, dec_layer=6, embed_dim=d, intermediate_dim=ff, use_obj=False,
                        num_attention_heads=heads, image_feat_dim=2048)
    m.set_compute_dtype(torch.bfloat16)
    m = m.to(dev).train()
    opt = AdamInverseSqrtWithWarmup(m.parameters(), lr=1e-4, betas=(0.9, 0.98), warmup_updates=4000)
    g = torch.Generator().manual_seed(1234)
    feats = torch.randn(32, 49, 2048, generator=g).to(dev)
    cap = torch.randint(6, V, (32, 32), generator=g); cap[:, 0] = 6; cap[:, -1] = 4
    cap = cap.to(dev)
    langs = torch.ones(32, dtype=torch.long)

    def step():
        loss, n = m.loss_fused(batch={"images": feats}, tgt_inputs=cap, tgt_langs=langs, tgt_mask=cap != 0, pad_idx=0)
        loss.backward()
        opt.step(max_grad_norm=1.0, zero_grad=True)
        return n
    sec, n = timed(step)
    return {"workload": "C3 captioning: feats [32,49,2048] -> 6L decoder d=512, captions [32,32]", "ms_per_step": round(1e3 * sec, 3),
            "tokens_per_s": round(n / sec, 1), "algorithmic_tflop_per_step": 0.265, "tflops": round(0.265 / sec, 1)}


def c4():
    torch.manual_seed(1234)
    m = ImageMassSeq2Seq(tp, lang_dec=False, enc_layer=6, dec_layer=6, embed_dim=d, intermediate_dim=ff, num_attention_heads=heads)
    m.set_compute_dtype(torch.bfloat16)
    m = m.to(dev).train()
    opt = AdamInverseSqrtWithWarmup(m.parameters(), lr=1e-4, betas=(0.9, 0.98), warmup_updates=4000)
    g = torch.Generator().manual_seed(1234)
    src = torch.randint(6, V, (64, 256), generator=g); src[:, 0] = 5; src[:, -1] = 4
    pad_idx = torch.full((64,), 255, dtype=torch.long)
    langs = torch.zeros(64, dtype=torch.long)
    src_dev = src.to(dev)

    def step():
        masked = mass_mask_device(0.3, pad_idx, src_dev.clone(), tp, seed=random.getrandbits(62))
        loss, n = m.loss_fused(src_inputs=masked["src_text"], tgt_inputs=masked["to_recover"], src_langs=langs, pad_idx=0,
                               tgt_positions=masked["positions"])
        loss.backward()
        opt.step(max_grad_norm=1.0, zero_grad=True)
        return n
    sec, n = timed(step)
    return {"workload": "C4 MASS: src [64,256], 127-token span masked on device, 6L/6L d=512", "ms_per_step": round(1e3 * sec, 3),
            "tokens_per_s": round(n / sec, 1), "algorithmic_tflop_per_step": 4.273, "tflops": round(4.273 / sec, 1)}


if __name__ == "__main__":
    random.seed(0)
    for fn in (c3, c4):
        print(json.dumps(fn()), flush=True)
