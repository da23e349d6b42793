"""N > 1 path with the REAL kernels: two processes share the one GPU of the box and exchange gradients through
`gloo` (RCCL needs one device per rank), so the layer-by-layer backward segments, the bucket hooks, the 1/world
scaling and the fused clip+Adam run exactly as in a multi-GPU job.  Both ranks must end with identical parameters,
equal to a single-process step on the averaged gradients of the two batches."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["encoder.encoder.layer.0.attention.self.query.weight", "decoder.decoder.layer.1.crossattention.self.key.weight",
        "output_layer.1.layer.weight", "output_layer.1.layer.bias", "encoder.embeddings.word_embeddings.weight",
        "decoder.decoder.layer.0.output.LayerNorm.weight"]


def _batch(seed, B=6, S=24, T=20, V=500):
    g = torch.Generator().manual_seed(seed)
    src, tgt = torch.randint(6, V, (B, S), generator=g), torch.randint(6, V, (B, T), generator=g)
    ls, lt = torch.randint(S // 2, S + 1, (B,), generator=g), torch.randint(T // 2, T + 1, (B,), generator=g)
    src[torch.arange(S)[None] >= ls[:, None]] = 0
    tgt[torch.arange(T)[None] >= lt[:, None]] = 0
    return {"src_texts": src, "dst_texts": tgt, "src_pad_mask": src != 0, "dst_pad_mask": tgt != 0,
            "src_langs": torch.zeros(B, dtype=torch.long), "dst_langs": torch.ones(B, dtype=torch.long)}


def _model():
    from imagetranslate_amd.seq2seq import Seq2Seq
    from imagetranslate_amd.textprocessor import SyntheticTextProcessor
    torch.manual_seed(7)
    return Seq2Seq(SyntheticTextProcessor(500), lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=128, intermediate_dim=256,
                   num_attention_heads=4).cuda().eval()  # eval: no dropout, so the two ways of computing agree tightly


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imagetranslate_amd.parallel import GradSync, train_step
        from imagetranslate_amd.utils import build_optimizer
        torch.cuda.set_device(0)
        m = _model()
        sync = GradSync(m, bucket_bytes=256 << 10)
        opt = build_optimizer(m, 1e-3, 4)
        losses = []
        for step in range(2):
            loss, ntok = train_step(m, opt, _batch(100 + 10 * step + rank), sync=sync, clip=1.0)
            losses.append(float(loss.detach()))
            assert len(sync.launched_buckets) >= 3, sync.launched_buckets
        named = dict(m.named_parameters())
        q.put((rank, "ok", losses, {k: named[k].detach().float().cpu().numpy() for k in KEYS}))  # numpy: pickled by value
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc()), None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_on_one_gpu_match_single_process(cuda):
    from imagetranslate_amd.utils import build_optimizer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]
    for r in res:
        for k in KEYS:
            r[3][k] = torch.from_numpy(r[3][k])
    for k in KEYS:
        assert torch.equal(res[0][3][k], res[1][3][k]), "ranks diverged on " + k
    # single process: accumulate both ranks' gradients, scale by 1/2 inside the fused optimizer step
    m = _model()
    opt = build_optimizer(m, 1e-3, 4)
    for step in range(2):
        for rank in range(2):
            b = _batch(100 + 10 * step + rank)
            loss, _ = m.loss_fused(b["src_texts"], b["dst_texts"], b["src_pad_mask"], b["dst_pad_mask"], b["src_langs"], b["dst_langs"])
            loss.backward()
            assert float(loss.detach()) == pytest.approx(res[rank][2][step], rel=1e-5)
        opt.step(max_grad_norm=1.0, grad_scale=0.5, zero_grad=True)
    named = dict(m.named_parameters())
    for k in KEYS:
        a, b = named[k].detach().float().cpu(), res[0][3][k]
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), k


def test_rccl_through_the_c_abi_single_rank(cuda):
    """imt_comm_* (RCCL opened by the library itself) with the one GPU of the test box: unique id -> communicator of one rank ->
    in-place all-reduce (identity for one rank) and broadcast on the communicator's side stream, ordered against the caller's
    stream by events; then GradSync driving its static bucket schedule through it."""
    import ctypes
    from imagetranslate_amd import _lib as L
    from imagetranslate_amd.parallel import GradSync, RcclComm
    from imagetranslate_amd.param_store import store_of
    lib = L.load()
    assert lib.imt_comm_unique_id_bytes() == 128
    comm = RcclComm(0, 1)
    x = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    y = x * 2.0                      # producer on the current stream
    w = comm.all_reduce_async(y)     # must run after it
    w.wait()
    z = y + 1.0                      # consumer on the current stream: after the collective
    torch.cuda.synchronize()
    assert torch.equal(z, x * 2.0 + 1.0)
    b = torch.randn(4096, device="cuda").bfloat16()
    ref = b.clone()
    comm.all_reduce_async(b).wait()
    comm.broadcast(b, 0)
    torch.cuda.synchronize()
    assert torch.equal(b, ref)
    # argument checks
    assert lib.imt_comm_allreduce(None, None, 4, 0, None) == -1
    assert lib.imt_comm_init(None, 1, 0, None) == -1
    # GradSync over the communicator (world of one: forced through the exchange path by pretending two ranks' schedule)
    m = _model()
    sync = GradSync(m, comm=comm, bucket_bytes=256 << 10)
    st = store_of(m.encoder)
    sync.world_size = 2  # drive the schedule; the communicator itself has one rank, so sums are identities
    sync.begin_step()
    st.grad.fill_(1.5)
    sync.output_layers_done()
    for mod, l in [(m.decoder, 1), (m.decoder, 0), (m.encoder, 1), (m.encoder, 0)]:
        sync._on_segment(mod, l)
    scale = sync.finish()
    torch.cuda.synchronize()
    assert scale == 0.5 and sync.launched_buckets == sync.bucket_schedule(None) and bool((st.grad == 1.5).all())
    lib.imt_set_gemm_share_cus(0)
    comm.destroy()


def test_bench_two_ranks_through_the_plain_command_line(cuda):
    """`python bench.py --gpus 2` with no launcher around it: the script starts its own two rank processes.  On the one-GPU
    test box both ranks share cuda:0 and gloo carries the collectives (IMT_BENCH_SINGLE_DEVICE / IMT_BENCH_BACKEND, rehearsal
    knobs); everything else -- per-rank batches, the static bucket schedule with the idle head left out, max-over-ranks timing,
    rank 0's one JSON line -- is the N > 1 path the driver runs on an 8-GPU node."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(IMT_BENCH_BACKEND="gloo", IMT_BENCH_SINGLE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "toy", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["global_batch"] == 16 and out["value"] > 0
    full = out["config"]["grad_exchange_bytes_per_step"]
    assert 0 < full  # the idle language head is left out of the exchange
