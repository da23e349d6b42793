#!/usr/bin/env python3
"""Two half-batch encoder forwards on two HIP streams against one full-batch forward (the stack runs inside one C call,
so the host is not the limit): do the launch boundaries of one stream fill with the other stream's kernels?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, build_model, make_batch  # noqa: E402

c = dict(CONFIGS["c1"])
model = build_model(c, torch.bfloat16, torch.device("cuda")).eval()
b = make_batch(c, 1, "cuda")
src, mask, langs = b["src_texts"], b["src_pad_mask"], b["src_langs"]


def enc(sl):
    with torch.no_grad():
        return model.encode(src[sl], mask[sl], langs[sl])[0]


def run(parts, streams, reps=20):
    for _ in range(3):
        for s, sl in zip(streams, parts):
            with torch.cuda.stream(s):
                enc(sl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for s, sl in zip(streams, parts):
            with torch.cuda.stream(s):
                enc(sl)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
full, h1, h2 = slice(0, 64), slice(0, 32), slice(32, 64)
print("full batch, one stream           : %.3f ms per encoder forward" % run([full], [s1]))
print("half batch, one stream           : %.3f ms" % run([h1], [s1]))
print("two halves, one stream           : %.3f ms" % run([h1, h2], [s1, s1]))
print("two halves, two streams          : %.3f ms" % run([h1, h2], [s1, s2]))
