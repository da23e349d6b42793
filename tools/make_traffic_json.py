#!/usr/bin/env python3
"""gpurun_out/pmc_traffic.json (per kernel symbol, written by tools/collect_traffic.sh) -> profiles/r<NN>_pmc_traffic.json (round = argv[2], default 2)
with a `per_kind` table keyed by the profiler kinds bench.py reports (gemm_<variant>_<dtype>_<layout>, ...)."""
import json, re, sys

raw = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_traffic.json"))
ROUND = int(sys.argv[2]) if len(sys.argv) > 2 else 2
VAR = {"gemm_kernel": "dbuf", "gemm_pipe_kernel": "dma", "gemm_sb_kernel": "sbuf", "gemm_ws_kernel": "ws", "gemm_xl_kernel": "xl"}
LAY = {"0": "nt", "1": "nn", "2": "tn"}
OTHER = [("gemm_grouped_tn_kernel", "gemm_bf16_tn_grouped"), ("attn_bwd_fused_kernel", "attn_bwd_fused_bf16"), ("attn_fwd_kernel", "attn_fwd_bf16"), ("attn_fwd_short_kernel", "attn_fwd_bf16"),
         ("ln_bwd_kernel", "layernorm_bwd"), ("ln_fwd_kernel", "layernorm_fwd"), ("gemm_ln_kernel", "gemm_ln_bf16"), ("clip_adam_kernel", "clip_adam"),
         ("xent_fused", "xent_fused"), ("sumsq_kernel", "grad_sumsq"), ("embed_bwd_kernel", "embed_bwd"), ("embed_fwd_kernel", "embed_fwd")]

def kind_of(sym):
    for k, var in VAR.items():
        if k in sym and "grouped" not in sym:
            m = re.search(k + r"I(DF16b|f)Li([012])", sym)  # mangled: element type and layout are visible
            if m:
                return "gemm_%s_%s_%s" % (var, "bf16" if m.group(1) == "DF16b" else "f32", LAY[m.group(2)])
            # the demangler prints the bf16 / layout-1 instantiation as "<bool _Accum, int, E>": that is bf16 NN
            return "gemm_%s_bf16_nn" % var
    for k, kind in OTHER:
        if k in sym:
            return kind
    return None

per_kind = {}
for sym, v in raw.items():
    k = kind_of(sym)
    if k is None:
        continue
    e = per_kind.setdefault(k, {"launches_profiled": 0, "_bytes": 0.0})
    e["launches_profiled"] += v["launches"]
    e["_bytes"] += v["hbm_bytes_avg"] * v["launches"]
for e in per_kind.values():
    e["hbm_bytes_per_launch"] = e.pop("_bytes") / e["launches_profiled"]
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/collect_traffic.sh) on `bench.py --steps 2 --warmup 1`; "
                 "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE reports half of a wide "
                 "coalesced read stream; the fabric counters include Infinity-Cache hits)",
       "round": ROUND, "per_kind": per_kind, "per_kernel": raw}
json.dump(out, open("profiles/r%02d_pmc_traffic.json" % ROUND, "w"), indent=1)
for k, v in sorted(per_kind.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_profiled"]):
    print("%-24s n=%4d %9.1f MB/launch" % (k, v["launches_profiled"], v["hbm_bytes_per_launch"] / 1e6))
