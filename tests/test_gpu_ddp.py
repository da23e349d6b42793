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
