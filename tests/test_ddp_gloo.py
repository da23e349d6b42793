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
