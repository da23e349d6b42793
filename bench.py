#!/usr/bin/env python3
"""Benchmark of the hot path: BASELINE.json's metric -- train tokens/sec (whole node) of the 6L enc-dec d=512
seq128 b64 MT train step -- on synthetic random-token batches.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., or
     plainly: without WORLD_SIZE in the environment the script starts its own N rank processes, one per GPU, BEFORE
     anything touches the GPU, and exits with the first non-zero rank exit code.)

The timed loop rotates through 8 distinct synthetic batches and alternates the two language directions, so neither
cache residency of one batch nor the optimizer's skip of never-touched embedding rows / of the idle language head
flatters the number.

One "step" = one pass of the hot path over one batch per rank: forward, label-smoothed NLL, backward, RCCL
gradient all-reduce (N > 1, overlapped), global grad-norm clip, Adam -- the body of ImageMTTrainer.train_epoch
(src/train_image_mt.py:239-295).  Tokens = non-pad TARGET tokens, the reference's own count (:256,:302-306).
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0      # HBM3E spec (6.3 TB/s is what a float4 copy reaches, same guide)
HBM_KINDS = ("layernorm_fwd", "layernorm_bwd", "embed_ln_fwd", "embed_fwd", "embed_bwd", "clip_adam", "xent_fused", "grad_sumsq")
TRAFFIC_FILE = os.path.join("profiles", "r03_pmc_traffic.json")

CONFIGS = {
    # name: (B, S, T, d, heads, ff, enc, dec, V)
    "c1": dict(B=64, S=128, T=128, d=512, heads=8, ff=2048, enc=6, dec=6, V=30000,
               desc="6L/6L enc-dec d=512 h=8 ff=2048 V=30000, src[64,128] tgt[64,128], synthetic random-token MT"),
    "toy": dict(B=8, S=32, T=32, d=128, heads=4, ff=512, enc=2, dec=2, V=1000,
                desc="2L/2L d=128 h=4 ff=512 V=1000 (plumbing)"),
    # side measurements (never the headline): SURVEY 8(d)'s paper-size vocabulary, its ragged-length variant of C1, and
    # the reference's own default model size (src/seq2seq.py:21-23: d=768, ff=3072, 12 heads, 6 encoder / 3 decoder layers)
    "c1v60k": dict(B=64, S=128, T=128, d=512, heads=8, ff=2048, enc=6, dec=6, V=60000,
                   desc="C1 with the paper's 60k vocabulary"),
    "c1ragged": dict(B=64, S=128, T=128, d=512, heads=8, ff=2048, enc=6, dec=6, V=30000, ragged=True,
                     desc="C1 with sentence lengths ~ U[64,128] (padding: key masks, non-pad row selection)"),
    "ref768": dict(B=64, S=128, T=128, d=768, heads=12, ff=3072, enc=6, dec=3, V=30000,
                   desc="reference default size: 6L/3L d=768 h=12 ff=3072 V=30000, src/tgt [64,128]"),
}


def algorithmic_flops(c):
    """2*M*N*K per GEMM (forward), x3 for a train step; SURVEY section 8(d) counting (masked positions dense)."""
    B, S, T1, d, ff, V, h = c["B"], c["S"], c["T"] - 1, c["d"], c["ff"], c["V"], c["heads"]
    Ns, Nt = B * S, B * T1
    enc = c["enc"] * (2 * Ns * d * 3 * d + 2 * Ns * d * d + 4 * B * S * S * d + 4 * Ns * d * ff)
    dec = c["dec"] * (2 * Nt * d * 3 * d + 2 * Nt * d * d + 4 * B * T1 * T1 * d            # self
                      + 2 * Nt * d * d + 2 * Ns * d * 2 * d + 2 * Nt * d * d + 4 * B * T1 * S * d  # cross
                      + 4 * Nt * d * ff)
    out = 2 * Nt * d * V
    return 3.0 * (enc + dec + out)


N_BATCHES = 8  # distinct batches the timed loop rotates through


def make_batch(c, seed, device, direction=0):
    """direction 0: language 0 -> 1, direction 1: language 1 -> 0 (the reference's MT data holds both, README.md:212-216)."""
    g = torch.Generator().manual_seed(seed)
    B, S, T, V = c["B"], c["S"], c["T"], c["V"]
    src = torch.randint(6, V, (B, S), generator=g)
    tgt = torch.randint(6, V, (B, T), generator=g)
    src[:, 0], tgt[:, 0] = (5, 6) if direction == 0 else (6, 5)   # language tags first (textprocessor.py:29-30), </s> last
    src[:, -1], tgt[:, -1] = 4, 4
    if c.get("ragged"):
        for x, L in ((src, S), (tgt, T)):
            lens = torch.randint(L // 2, L + 1, (B,), generator=g)
            for i in range(B):
                x[i, lens[i] - 1] = 4
                x[i, lens[i]:] = 0
    b = {"src_texts": src, "dst_texts": tgt, "src_pad_mask": src != 0, "dst_pad_mask": tgt != 0,
         "src_langs": torch.full((B,), direction, dtype=torch.long),
         "dst_langs": torch.full((B,), 1 - direction, dtype=torch.long)}
    out = {k: v.to(device) if k not in ("src_langs", "dst_langs") else v for k, v in b.items()}
    # the loader's host-side count of non-pad target positions (the reference: train_image_mt.py:253-256); with it the step
    # is enqueued without a device->host read.  IMT_BENCH_NO_COUNT=1: let the step read it back from the device instead.
    if not os.environ.get("IMT_BENCH_NO_COUNT"):
        out["ntokens"] = int(b["dst_pad_mask"][:, 1:].sum())
    return out


def build_model(c, dtype, device):
    from imagetranslate_amd.seq2seq import Seq2Seq
    from imagetranslate_amd.textprocessor import SyntheticTextProcessor
    tp = SyntheticTextProcessor(c["V"])
    torch.manual_seed(1234)
    m = Seq2Seq(tp, lang_dec=False, enc_layer=c["enc"], dec_layer=c["dec"], embed_dim=c["d"], intermediate_dim=c["ff"],
                num_attention_heads=c["heads"])
    m.set_compute_dtype(dtype)
    return m.to(device)


def cpu_baseline(c, seconds_budget=15.0):
    """Reference-equivalent CPU path (this repo's oracle, kind 'port') timed on the host cores on a bounded sample
    of the same workload: the C1 model with a batch of B/4 = 16 of the 64 sentences, full train step, on at most 16
    host threads."""
    from oracle import reference_model as R
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # a 1-GPU box grants a 16-CPU share whatever the host's core count (256 threads thrash: 240 s/step measured)
    cores = max(1, min(16, cores))
    torch.set_num_threads(cores)
    print("[bench] cpu_baseline: oracle train step on %d host threads ..." % cores, file=sys.stderr, flush=True)
    Bs = max(1, c["B"] // 4)
    tp = R.SyntheticTextProcessor(c["V"])
    torch.manual_seed(1234)
    m = R.Seq2Seq(tp, lang_dec=False, enc_layer=c["enc"], dec_layer=c["dec"], embed_dim=c["d"], intermediate_dim=c["ff"],
                  num_attention_heads=c["heads"]).train()
    opt = R.AdamInverseSqrtWithWarmup(m.parameters(), lr=1e-4, betas=(0.9, 0.98), warmup_updates=4000)
    crit = R.SmoothedNLLLoss(ignore_index=0)
    cs = dict(c, B=Bs)
    b = make_batch(cs, 1234, "cpu")
    tw = time.time()
    R.train_step(m, opt, crit, b)  # warm-up
    print("[bench] cpu_baseline: warm-up step %.1f s" % (time.time() - tw), file=sys.stderr, flush=True)
    t0, n, toks = time.time(), 0, 0
    while n < 8 and (time.time() - t0) < seconds_budget:
        _, nt = R.train_step(m, opt, crit, b)
        toks += nt
        n += 1
        print("[bench] cpu_baseline: step %d done at %.1f s" % (n, time.time() - t0), file=sys.stderr, flush=True)
    dt = time.time() - t0
    return {"value": toks / dt, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d full train steps (fwd+loss+bwd+clip+Adam) of the C1 model on %d of the 64 sentences "
                      "([%d,%d] src/tgt), fp32 torch eager, dropout 0.1 (oracle/reference_model.py)" % (n, Bs, Bs, c["S"])}


def profile_pass(step_fn, steps=2):
    """Re-run a few steps with the library's per-launch HIP-event profiler on; returns per-kind rows."""
    from imagetranslate_amd import _lib as L
    lib = L.load()
    torch.cuda.synchronize()
    lib.imt_prof_enable(1)
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    rows = (L.ProfRow * 256)()
    n = lib.imt_prof_report(rows, 256)
    lib.imt_prof_enable(0)
    out = []
    for i in range(n):
        r = rows[i]
        out.append({"kind": r.kind.decode(), "launches": int(r.launches) // steps, "ms": r.total_ms / steps,
                    "flops": r.flops / steps, "bytes": r.bytes / steps})
    return sorted(out, key=lambda r: -r["ms"])


def tune_dp_policy(sync, step, fence, device, backend, probe_steps=3):
    """Data-parallel run: two choices that cannot be made on a one-GPU box are made HERE by measurement, before the warm-up,
    identically on every rank (times are max-reduced over the ranks, so every rank sees the same numbers and picks the same
    minimum): (1) whether the one-tile-per-CU GEMMs with K < 1024 leave the persistent kernel for the three-workgroups-per-CU
    one while RCCL's kernels hold CUs (imt_set_gemm_share_cus); (2) whether the gradient buckets travel
    through torch.distributed's nccl backend or through the library's own RCCL communicator (imt_comm_*).  Each candidate
    runs `probe_steps` real train steps.  An axis pinned by its environment variable (IMT_GEMM_SHARE_CUS / IMT_COMM) is
    not tuned; the communicator axis is probed only with IMT_BENCH_TUNE_COMM=1 (see below).  Returns the choice and the probe times for the JSON line."""
    import torch.distributed as dist
    from imagetranslate_amd import _lib as L
    from imagetranslate_amd.parallel import RcclComm
    lib = L.load()
    share_axis = [None] if os.environ.get("IMT_GEMM_SHARE_CUS") is not None else [1, 0]
    comm_axis = ["torch-" + backend]
    rccl = None
    # The communicator axis is opt-in (IMT_BENCH_TUNE_COMM=1): imt_comm_* with more than one rank has never run on hardware
    # (one-GPU boxes only), and a collective that misbehaves inside the benchmark would take the whole scaling run with it;
    # the GEMM policy axis is a process-local switch and is always probed.
    if os.environ.get("IMT_BENCH_TUNE_COMM") == "1" and os.environ.get("IMT_COMM") is None and backend == "nccl" and sync.comm is None:
        ok = torch.ones(1, device=device)
        try:
            rccl = RcclComm(dist.get_rank(), dist.get_world_size())
        except Exception as err:  # librccl missing, ...: every rank must agree before anything is switched
            print("[bench] imt_comm unavailable on this rank: %r" % (err,), file=sys.stderr, flush=True)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok) > 0:
            comm_axis.append("imt-rccl")
        else:
            rccl = None
    elif sync.comm is not None:
        comm_axis = ["imt-rccl"]
        rccl = sync.comm
    probes = {}
    for comm in comm_axis:
        sync.comm = rccl if comm == "imt-rccl" else None
        for share in share_axis:
            if share is not None:
                lib.imt_set_gemm_share_cus(share)
            step()  # one untimed step per candidate (first use of a communicator / kernel variant)
            fence()
            t0 = time.perf_counter()
            for _ in range(probe_steps):
                step()
            fence()
            t = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            probes[(comm, share)] = 1e3 * float(t) / probe_steps
    best = min(probes, key=lambda k: (probes[k], k[0], -1 if k[1] is None else k[1]))
    sync.comm = rccl if best[0] == "imt-rccl" else None
    if best[1] is not None:
        lib.imt_set_gemm_share_cus(best[1])
    return {"comm": best[0], "comm_axis_probed": len(comm_axis) > 1, "gemm_share_cus": "env" if best[1] is None else best[1],
            "probe_ms_per_step": {"%s,share_cus=%s" % k: round(v, 3) for k, v in probes.items()}}


def self_launch(n):
    """python bench.py --gpus N without a launcher: start N rank processes of this script (nothing here has touched the
    GPU yet -- children are started, never exec'd into), wait for all, propagate the first failure."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print("[bench] rank %d exited with code %d: stopping the other ranks" % (r, code), file=sys.stderr, flush=True)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c1", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (args.gpus, world))
    import torch.distributed as dist
    # rehearsal knobs (NOT for measurements): IMT_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and
    # IMT_BENCH_BACKEND=gloo carries the collectives, so the N > 1 control flow can be exercised on a one-GPU box
    if os.environ.get("IMT_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("IMT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from imagetranslate_amd.parallel import GradSync, train_step
    from imagetranslate_amd.utils import AdamInverseSqrtWithWarmup

    c = CONFIGS[args.config]
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = build_model(c, dtype, device)
    model.train(not args.no_dropout)  # reference trains with dropout 0.1 (lm_config.py:6,8)
    opt = AdamInverseSqrtWithWarmup(model.parameters(), lr=1e-4, betas=(0.9, 0.98), warmup_updates=4000)
    # each rank its own batches (weak scaling: global batch = 64 * N); batch i of every rank has direction i % 2, so all
    # ranks use the same vocabulary projection in a step and the other one stays out of the gradient exchange
    batches = [make_batch(c, 1234 + 1000 * i + rank, device, direction=i % 2) for i in range(N_BATCHES)]
    if os.environ.get("IMT_BENCH_ONE_BATCH") == "1":  # round-1 behaviour, for comparisons only
        batches = batches[:1]
    sync = GradSync(model) if world > 1 else None
    ntoks = [int(b["dst_pad_mask"][:, 1:].sum()) for b in batches]
    counter = [0]

    def step():
        i = counter[0] % len(batches)
        counter[0] += 1
        return train_step(model, opt, batches[i], sync=sync, clip=1.0, active_head=int(batches[i]["dst_langs"][0]))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dp_policy = tune_dp_policy(sync, step, fence, device, backend) if sync is not None else None
    for _ in range(args.warmup):
        step()
    fence()
    first = counter[0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    fence()
    ntok_timed = sum(ntoks[(first + k) % len(batches)] for k in range(args.steps))
    ntok_step = ntok_timed / max(1, args.steps)
    elapsed = time.perf_counter() - t0
    ntok_all = ntok_timed
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        t = torch.tensor([float(ntok_timed)], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ntok_all = float(t)
    loss_val = float(loss.detach())

    # instrumented pass: EVERY rank runs the same extra steps (they contain collectives); only rank 0 records events
    if rank == 0:
        rows = profile_pass(step)
    else:
        rows = []
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = ntok_all / elapsed   # non-pad target tokens of ALL ranks in the timed steps / max-over-ranks time
        flops = algorithmic_flops(c)
        peak = PEAK_BF16_TFLOPS if dtype == torch.bfloat16 else PEAK_F32_TFLOPS
        # dominant kernel = the kernel kind with the largest summed device time in the instrumented pass
        gemm_rows = [r for r in rows if r["kind"].startswith("gemm_") and " " not in r["kind"]]
        dom = max(gemm_rows, key=lambda r: r["ms"]) if gemm_rows else None
        roofline = None
        if dom is not None:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            traffic, traffic_source = None, None
            for tf in (TRAFFIC_FILE, os.path.join("profiles", "r02_pmc_traffic.json")):
                try:  # HBM bytes per launch of this kernel kind: a REPLAY of the committed PMC passes (tools/collect_traffic.sh),
                      # not a measurement of this run -- PMC collection needs rocprofv3 around the process
                    pt = json.load(open(os.path.join(ROOT, tf)))
                    if args.config == "c1" and world == 1:
                        traffic = round(pt["per_kind"][dom["kind"]]["hbm_bytes_per_launch"])
                        traffic_source = tf
                        break
                except Exception:
                    traffic = None
            roofline = {"bound": "mfma", "kernel": dom["kind"], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                        "algorithmic_bytes_per_launch": round(dom["bytes"] / max(1, dom["launches"])),
                        "launches_per_step": dom["launches"], "avg_launch_us": round(1e3 * dom["ms"] / max(1, dom["launches"]), 2),
                        "step_frac": round(flops / (ms_per_step * 1e-3) / 1e12 / peak, 4),
                        "step_achieved": round(flops / (ms_per_step * 1e-3) / 1e12, 2)}
        if roofline is not None:
            # the HBM side of the roofline (north_star: achieved HBM GB/s against chip peak): every HBM-bound kernel kind of the
            # step, algorithmic bytes per launch / event-bracketed launch time of the instrumented pass
            hbm = []
            for r in rows:
                if r["kind"] in HBM_KINDS and r["launches"] > 0 and r["ms"] > 0:
                    gbps = r["bytes"] / (r["ms"] * 1e-3) / 1e9
                    hbm.append({"kernel": r["kind"], "launches_per_step": r["launches"], "bytes_per_launch": round(r["bytes"] / r["launches"]),
                                "avg_launch_us": round(1e3 * r["ms"] / r["launches"], 2), "achieved_GBps": round(gbps, 1),
                                "frac_of_8TBps": round(gbps / PEAK_HBM_GBPS, 4)})
            roofline["hbm_kernels"] = hbm
        out = {
            "metric": "train tokens/sec (whole node), 6L enc-dec d=512 seq128 b64, 1/2/4/8 GPU",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": c["desc"], "global_batch": c["B"] * world, "seq_len": c["S"],
                       "target_tokens_per_step": round(ntok_step * world, 1), "parallelism": "dp%d" % world,
                       "distinct_batches": len(batches), "directions": len({int(b["dst_langs"][0]) for b in batches}),
                       "dropout": 0.0 if args.no_dropout else 0.1, "algorithmic_tflop_per_step": round(flops / 1e12, 3),
                       "final_loss": round(loss_val, 4)},
            "roofline": roofline,
        }
        adam = [r for r in rows if r["kind"] == "clip_adam"]
        if adam:  # bytes the optimizer pass is charged per step (34 B per parameter element: nothing is skipped in the count)
            out["config"]["adam_bytes_per_step"] = round(sum(r["bytes"] for r in adam))
        if sync is not None:
            out["config"]["grad_exchange_bytes_per_step"] = sync.exchanged_bytes(0)
            out["config"]["dp_policy"] = dp_policy
        print("[bench] gpu: %.1f tokens/s, %.3f ms/step" % (value, ms_per_step), file=sys.stderr, flush=True)
        if args.breakdown:
            tot = sum(r["ms"] for r in rows)
            print("%-40s %8s %10s %9s %9s" % ("kernel", "launches", "ms/step", "TFLOP/s", "GB/s"), file=sys.stderr)
            for r in rows:
                print("%-40s %8d %10.3f %9.1f %9.1f" % (r["kind"], r["launches"], r["ms"], r["flops"] / (r["ms"] * 1e9 + 1e-30),
                                                        r["bytes"] / (r["ms"] * 1e6 + 1e-30)), file=sys.stderr)
            print("sum of kernel time %.3f ms / step (instrumented pass)" % tot, file=sys.stderr, flush=True)
        out["cpu_baseline"] = cpu_baseline(c) if (world == 1 and not args.no_cpu_baseline) else None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
