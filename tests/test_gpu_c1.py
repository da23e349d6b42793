"""Parity at the BENCHMARKED size (BASELINE.json configs[1], "C1": 6L/6L d=512 h=8 ff=2048 V=30000, T=128) against the
CPU oracle: the kernels and the shape-dependent dispatch that bench.py times -- the persistent wave-specialised GEMM, the
256-tile kernel, the grouped per-layer weight-gradient launch, the vocabulary-dX split-K slab branch (seq2seq.py: V >= 16384
and n >= 2048 rows) and the LDS-resident fused cross-entropy at 60-KB rows -- are reached only at this size, so they are
checked here, through the module API, on the same weights and batches as the oracle.

The FULL C1 batch is used (B = 64: 8128 target rows; 8128 = 127 * 64 is also what lets the decoder's weight gradients take
the grouped launch, whose K -- the token count -- must be a whole number of 64-token tiles); the oracle's forward + backward
of it takes a few seconds on the box's host cores, and is computed once per batch variant.
Tolerances: fp32 compute mode 1e-4 relative on log-probs AND on gradients through the 12-layer stack (north_star:
"fp32 logits and grads within 1e-4 relative"), argmax bit-exact.  Rounds 1-2 held the gradients to 3e-4 on an asserted
argument about summation orders; round 3 measured it against an fp64 run of the oracle (test_c1_fp32_against_fp64_truth,
profiles/r03_fp64_truth_c1.txt): over all 190 gradient tensors the HIP fp32 path is within 1.9e-5 of the truth and the fp32
oracle within 1.6e-5, so 1e-4 holds with a 3x margin for HIP-vs-oracle as well;
bf16 mode (the benchmarked arithmetic, bf16 storage + fp32 accumulate) 4e-2 on log-probs, 1e-1 on gradients."""
import copy
import ctypes
import os

import pytest
import torch

from oracle import reference_model as R
from tests.util import assert_close, fp64_truth_report

pytestmark = pytest.mark.gpu

C1 = dict(enc_layer=6, dec_layer=6, embed_dim=512, intermediate_dim=2048, num_attention_heads=8)
V = 30000
GRAD_KEYS = ["encoder.embeddings.word_embeddings.weight",
             "encoder.encoder.layer.0.attention.self.query.weight",           # shared with decoder layer 0 (src/seq2seq.py:63-65)
             "decoder.decoder.layer.3.crossattention.self.key.weight",        # batched cross K|V projection
             "decoder.decoder.layer.5.crossattention.self.value.bias",
             "encoder.encoder.layer.2.intermediate.dense.weight", "decoder.decoder.layer.1.output.dense.weight",
             "decoder.decoder.layer.4.output.LayerNorm.weight", "encoder.encoder.layer.5.attention.output.LayerNorm.bias",
             "output_layer.1.layer.weight", "output_layer.1.layer.bias"]


def _batch(B=64, S=128, T=128, seed=4321, ragged=False):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(6, V, (B, S), generator=g)
    tgt = torch.randint(6, V, (B, T), generator=g)
    src[:, 0], tgt[:, 0] = 5, 6
    src[:, -1], tgt[:, -1] = 4, 4
    if ragged:  # bench.py's c1ragged: lengths ~ U[64, 128]
        for x, L in ((src, S), (tgt, T)):
            lens = torch.randint(L // 2, L + 1, (B,), generator=g)
            lens[0] = L  # keep the full width
            for i in range(B):
                x[i, lens[i] - 1] = 4
                x[i, lens[i]:] = 0
    return (src, tgt, src != 0, tgt != 0, torch.zeros(B, dtype=torch.long), torch.ones(B, dtype=torch.long))


@pytest.fixture(scope="module")
def pair(cuda):
    from imagetranslate_amd.seq2seq import Seq2Seq
    torch.manual_seed(20)
    tp = R.SyntheticTextProcessor(V)
    ref = R.Seq2Seq(tp, lang_dec=False, **C1).eval()
    # N(0, 0.02) weights give near-uniform next-token distributions; scale the matrices so attention and the softmax over
    # the vocabulary are not degenerate (same trick as the beam-search fixtures)
    with torch.no_grad():
        for k, p in ref.named_parameters():
            if p.dim() > 1:
                p.mul_(2.0)
            elif k.endswith("bias"):
                p.normal_(0.0, 0.02)
    ours = Seq2Seq(tp, lang_dec=False, **C1)
    ours.load_state_dict(ref.state_dict())
    return ref, ours.cuda().eval()


_ORACLE = {}


def _oracle(ref, args, key=None):
    if key is not None and key in _ORACLE:
        return _ORACLE[key]
    out = _oracle_run(ref, args)
    if key is not None:
        _ORACLE[key] = out
    return out


def _oracle_run(ref, args):
    ref.zero_grad()
    lp = ref(*args, log_softmax=True)
    targets = args[1][:, 1:][args[3][:, 1:]]
    loss = R.SmoothedNLLLoss(ignore_index=0)(lp, targets).mean()
    loss.backward()
    return lp.detach(), float(loss.detach()), {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}


def _oracle64(ref, args):
    """The oracle in fp64 on the same weights and batch: ground truth for the 1e-4 bar (about a minute on the box's cores)."""
    ref64 = copy.deepcopy(ref).double()
    ref64.zero_grad()
    lp = ref64(*args, log_softmax=True)
    targets = args[1][:, 1:][args[3][:, 1:]]
    loss = R.SmoothedNLLLoss(ignore_index=0)(lp, targets).mean()
    loss.backward()
    return lp.detach(), float(loss.detach()), {k: p.grad for k, p in ref64.named_parameters() if p.grad is not None}


def write_truth_log(name, lines):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "fp64_truth_%s.txt" % name), "w") as fw:
            fw.write("\n".join(lines) + "\n")
    except OSError:
        pass


def _kinds_of_step(fn):
    """Kernel kinds (imt_prof_report rows) launched by fn()."""
    from imagetranslate_amd import _lib as L
    lib = L.load()
    torch.cuda.synchronize()
    lib.imt_prof_enable(1)
    try:
        out = fn()
        torch.cuda.synchronize()
        rows = (L.ProfRow * 256)()
        n = lib.imt_prof_report(rows, 256)
    finally:
        lib.imt_prof_enable(0)
    return out, {rows[i].kind.decode(): int(rows[i].launches) for i in range(n)}


@pytest.mark.parametrize("ragged", [False, True], ids=["c1", "c1ragged"])
def test_c1_fp32_parity_and_dispatch(pair, ragged):
    ref, ours = pair
    args = _batch(ragged=ragged)
    lp_ref, loss_ref, g_ref = _oracle(ref, args, ("c1", ragged))
    n_rows = int(args[3][:, 1:].sum())
    assert n_rows >= 2048 or ragged
    ours.set_compute_dtype(torch.float32)
    ours.zero_grad()
    with torch.no_grad():
        lp = ours(*args, log_softmax=True)
    assert lp.shape == (n_rows, V)
    assert_close(lp, lp_ref, 1e-4, "C1 fp32 log-probs")
    assert torch.equal(lp.argmax(-1).cpu(), lp_ref.argmax(-1)), "argmax token ids must be bit-exact"
    del lp

    def step():
        loss, ntok = ours.loss_fused(*args)
        loss.backward()
        return loss, ntok
    (loss, ntok), kinds = _kinds_of_step(step)
    assert ntok == n_rows
    assert abs(float(loss.detach()) - loss_ref) <= 1e-5 * abs(loss_ref), (float(loss.detach()), loss_ref)
    ours_g = dict(ours.named_parameters())
    for k in GRAD_KEYS:
        assert_close(ours_g[k].grad, g_ref[k], 1e-4, "C1 fp32 grad " + k)
    # the dispatch bench.py times
    assert kinds.get("gemm_ws_f32_nn", 0) + kinds.get("gemm_ws_f32_nt", 0) >= 30, kinds
    assert kinds.get("gemm_f32_tn_grouped", 0) == 12, kinds
    assert any(k.startswith("xent_fused") for k in kinds), kinds


def test_c1_fp32_against_fp64_truth(pair):
    """north_star: "fp32 logits and grads within 1e-4 relative".  Ground truth = the oracle run in fp64 on the full C1 batch;
    the HIP fp32 path must be within 1e-4 of it (max |a-t| / max |t|) on the log-probs and on EVERY parameter gradient
    (word / position / type embeddings through atomics, LayerNorm gains through the 32-copy partials, the vocabulary dX through
    the split-K slabs included), and within 2e-3 element-wise on the entries above 1 % of a tensor's maximum.  The fp32
    oracle's own distance from the truth is printed beside each number (gpurun_out/fp64_truth_c1.txt)."""
    ref, ours = pair
    args = _batch()
    lp32, loss32, g32 = _oracle(ref, args, ("c1", False))
    lp64, loss64, g64 = _oracle64(ref, args)
    ours.set_compute_dtype(torch.float32)
    log = []
    with torch.no_grad():
        lp = ours(*args, log_softmax=True)
    fp64_truth_report("log-probs [8128, 30000]", lp, lp32, lp64, 1e-4, 2e-3, log)
    assert torch.equal(lp.argmax(-1).cpu(), lp64.argmax(-1)), "argmax token ids must equal the fp64 truth's"
    del lp, lp64
    ours.zero_grad()
    loss, _ = ours.loss_fused(*args)
    loss.backward()
    assert abs(float(loss.detach()) - loss64) <= 1e-5 * abs(loss64), (float(loss.detach()), loss64, loss32)
    worst, worst_o = 0.0, 0.0
    n = 0
    try:
        for k, p in ours.named_parameters():
            if k not in g64 or p.grad is None:
                continue
            if float(g64[k].abs().max()) < 1e-9:  # zero in exact arithmetic (attention key biases)
                continue
            e, eo = fp64_truth_report("grad " + k, p.grad, g32[k], g64[k], 1e-4, 2e-3, log)
            worst, worst_o, n = max(worst, e), max(worst_o, eo), n + 1
    finally:
        log.append("tensors %d; worst hip-fp32 %.3e, worst oracle-fp32 %.3e" % (n, worst, worst_o))
        write_truth_log("c1", log)
    assert n >= 150, n


@pytest.mark.parametrize("ragged", [False, True], ids=["c1", "c1ragged"])
def test_c1_bf16_parity_and_dispatch(pair, ragged):
    """The benchmarked arithmetic.  Tolerances (bf16 has 8 significant bits; 12 layers): log-probs 4e-2 relative to the
    largest |log-prob|, loss 2e-2, gradients 1e-1 relative to the largest entry of each tensor."""
    ref, ours = pair
    args = _batch(ragged=ragged)
    lp_ref, loss_ref, g_ref = _oracle(ref, args, ("c1", ragged))
    ours.set_compute_dtype(torch.bfloat16)
    ours.zero_grad()
    with torch.no_grad():
        lp = ours(*args, log_softmax=True)
    assert_close(lp, lp_ref, 4e-2, "C1 bf16 log-probs")
    agree = float((lp.argmax(-1).cpu() == lp_ref.argmax(-1)).float().mean())
    assert agree > 0.97, "bf16 argmax agreement %.3f" % agree
    del lp

    def step():
        loss, ntok = ours.loss_fused(*args)
        loss.backward()
        return loss, ntok
    (loss, ntok), kinds = _kinds_of_step(step)
    assert abs(float(loss.detach()) - loss_ref) <= 2e-2 * abs(loss_ref), (float(loss.detach()), loss_ref)
    ours_g = dict(ours.named_parameters())
    for k in GRAD_KEYS:
        assert_close(ours_g[k].grad, g_ref[k], 1e-1, "C1 bf16 grad " + k)
    # every kernel family of the timed step ran: persistent wave-specialised GEMMs, 256-tile GEMMs (FFN up / vocabulary
    # projection / batched cross K|V), the grouped weight gradients, the split-K slab dX through the vocabulary, the fused
    # cross-entropy, fused attention
    assert sum(v for k, v in kinds.items() if k.startswith("gemm_ws_bf16")) >= 30, kinds
    assert sum(v for k, v in kinds.items() if k.startswith("gemm_xl_bf16")) >= 10, kinds
    assert kinds.get("gemm_bf16_tn_grouped", 0) == 12, kinds
    assert kinds.get("gemm_splitk_reduce", 0) == 1, kinds
    assert any(k.startswith("xent_fused") for k in kinds), kinds
    assert any(k.startswith("attn_bwd") for k in kinds) and any(k.startswith("attn_fwd") for k in kinds), kinds
    ours.set_compute_dtype(torch.float32)


def test_c1_bf16_train_mode_dropout_is_consistent(pair):
    """Train mode (dropout 0.1) at C1 size: the backward regenerates the forward's masks, so two runs with the same pinned
    seeds give bit-identical gradients, and the loss stays near the eval loss."""
    ref, ours = pair
    args = _batch(seed=777)
    ours.set_compute_dtype(torch.bfloat16)
    ours.train()
    for st in ours._stacks():
        st._imt_dropout_seed = 1234
    try:
        grads = []
        for _ in range(2):
            ours.zero_grad()
            loss, _ = ours.loss_fused(*args)
            loss.backward()
            torch.cuda.synchronize()
            grads.append({k: dict(ours.named_parameters())[k].grad.clone() for k in GRAD_KEYS[:4]})
        for k in grads[0]:
            # bit-equal for the GEMM-produced weight gradients; the embedding / LayerNorm gradients are fp32 atomic sums whose
            # order varies: 1e-5 (a mask mismatch between forward and backward would be an O(1) difference)
            if "embeddings" in k or "LayerNorm" in k:
                assert_close(grads[0][k], grads[1][k], 1e-5, "repeat " + k)
            else:
                assert torch.equal(grads[0][k], grads[1][k]), "dropout masks of backward differ from forward's: " + k
        assert torch.isfinite(loss).all()
    finally:
        for st in ours._stacks():
            st._imt_dropout_seed = None
        ours.eval()
        ours.set_compute_dtype(torch.float32)
