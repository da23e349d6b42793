"""Label-smoothed NLL on log-probabilities -- drop-in for the reference's ``src/loss.py:4-27``.

``SmoothedNLLLoss(ignore_index=pad)(log_probs[N,V], target[N]) -> [N,1]`` (``reduce`` is forced False exactly as
the reference does, ``loss.py:8``; the trainer takes ``.mean()``, ``src/train_image_mt.py:282``).  Forward and
backward are HIP kernels (``imt_smoothed_nll_fwd/bwd``); there is no CPU fallback.
"""
import torch
import torch.nn as nn

from . import hip_ops as O


class _SmoothedNLLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lp, target, epsilon, ignore_index):
        lp = O.rows16(lp.float())
        target = target.to(lp.device).contiguous()
        ctx.save_for_backward(target)
        ctx.V, ctx.eps, ctx.ign = lp.shape[1], epsilon, ignore_index
        return O.smoothed_nll_fwd(lp, target, epsilon, ignore_index)

    @staticmethod
    def backward(ctx, dloss):
        (target,) = ctx.saved_tensors
        return O.smoothed_nll_bwd(dloss.float().contiguous(), target, ctx.V, ctx.eps, ctx.ign), None, None, None


class SmoothedNLLLoss(nn.NLLLoss):
    def __init__(self, weight=None, ignore_index=-100, reduce: bool = False, epsilon=0.1):
        super().__init__(weight=weight, ignore_index=ignore_index)
        self.epsilon = epsilon
        self.reduce = False

    def forward(self, input, target):
        if target.dim() == input.dim():
            target = target.squeeze(-1)
        ign = self.ignore_index if self.ignore_index is not None else -(2 ** 62)
        return _SmoothedNLLFn.apply(input, target.reshape(-1), float(self.epsilon), int(ign))
