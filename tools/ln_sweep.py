#!/usr/bin/env python3
"""LayerNorm-backward block-count sweep (IMT_LN_RPW hook), atomics tail vs grouped last-block reduction."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagetranslate_amd import hip_ops as O
from tools.rowop_bench import timeit  # noqa

rows, d = 8192, 512
x = torch.randn(rows, d, device="cuda").bfloat16(); dy = torch.randn(rows, d, device="cuda").bfloat16()
g = torch.ones(d, device="cuda").bfloat16(); b = torch.zeros(d, device="cuda").bfloat16()
y, mean, rstd = O.layernorm_fwd(x, g, b)
dg = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
for wpb in (4, 8):
    for nblk in (256, 512, 1024, 2048):
        os.environ["IMT_LN_WPB"], os.environ["IMT_LN_BLOCKS"] = str(wpb), str(nblk)
        ta = timeit(lambda: O.layernorm_bwd(dy, x, g, mean, rstd, dg, db))
        td = timeit(lambda: O.layernorm_bwd(dy, x, g, mean, rstd, dg, db, want_dx_drop=True, dx_dropout_p=0.1, dx_dropout_seed=5))
        print("waves/block %2d blocks %4d: %6.1f us (%5.0f GB/s) | + dropout output %6.1f us" %
              (wpb, nblk, ta, 3 * rows * d * 2 / ta / 1e3, td), flush=True)
