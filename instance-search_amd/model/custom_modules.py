"""Descriptor-head ops with the reference's names (model/custom_modules.py):
NormalizeL2Fun / NormalizeL2 (:46-76) and ShiftFun / Shift (:11-39).

Forward on a GPU tensor runs the hand-written HIP kernel (libisx `isx_l2norm_rows`); it
raises if the library is missing -- there is no silent torch fallback on the GPU.  On a CPU
tensor (the reference's `--device=-1`, BASELINE config 1 "plumbing, no GPU") the same
arithmetic runs in plain torch.  Backward follows the reference's formula (:59-67) so the
modules stay usable in training graphs.  The losses (MetricLoss, TripletLoss :81-215) are
training-only and out of scope for this round."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.nn.parameter import Parameter

EPS = 1e-10


def l2_normalize_rows(x, eps=EPS):
    """y = x / sqrt(sum_j x_j^2 + eps): eps inside the sqrt, NOT max(norm, eps)."""
    if x.is_cuda:
        from isx import ops
        return ops.l2norm_rows(x.float(), eps)
    return x / (x.pow(2).sum(1, keepdim=True) + eps).sqrt()


class _NormalizeL2(Function):
    @staticmethod
    def forward(ctx, x, eps=EPS):
        y = l2_normalize_rows(x, eps)
        ctx.save_for_backward(x)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        norm2 = x.pow(2).sum(1, keepdim=True) + ctx.eps
        norm = norm2.sqrt()
        cross = (x * grad_output).sum(1, keepdim=True)
        grad = (norm2 * grad_output - x * cross) / (norm2 * norm)
        return grad, None


class NormalizeL2Fun(object):
    """`NormalizeL2Fun()(x)` -- the reference's legacy call form (train/classif_finetune.py:100) --
    and `NormalizeL2Fun.apply(x, eps)` both work; the autograd node is _NormalizeL2."""

    def __init__(self, eps=EPS):
        self.eps = eps

    def __call__(self, x):
        return _NormalizeL2.apply(x, self.eps)

    forward = __call__
    apply = staticmethod(lambda x, eps=EPS: _NormalizeL2.apply(x, eps))


class NormalizeL2(nn.Module):
    def forward(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            return NormalizeL2Fun.apply(x, EPS)
        return l2_normalize_rows(x, EPS)


class _Shift(Function):
    @staticmethod
    def forward(ctx, x, param):
        return x + param.view(1, -1)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, grad_output.sum(0)


class ShiftFun(object):
    def __call__(self, x, param):
        return _Shift.apply(x, param)

    forward = __call__
    apply = staticmethod(lambda x, param: _Shift.apply(x, param))


class Shift(nn.Module):
    """y = x + param (one trainable offset per feature, initialised to 0)."""

    def __init__(self, n_features):
        super().__init__()
        self.param = Parameter(torch.zeros(n_features))

    def reset_parameters(self):
        self.param.data.fill_(0)

    def forward(self, x):
        return x + self.param.view(1, -1)
