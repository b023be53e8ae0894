"""Descriptor-head ops with the reference's names (model/custom_modules.py):
NormalizeL2Fun / NormalizeL2 (:46-76) and ShiftFun / Shift (:11-39).

Forward on a GPU tensor runs the hand-written HIP kernel (libisx `isx_l2norm_rows`); it
raises if the library is missing -- there is no silent torch fallback on the GPU.  On a CPU
tensor (the reference's `--device=-1`, BASELINE config 1 "plumbing, no GPU") the same
arithmetic runs in plain torch.  Backward follows the reference's formula (:59-67) so the
modules stay usable in training graphs.  TripletLoss (:140-215) runs its per-row forward and its
analytic backward as HIP kernels on the GPU (`isx_triplet_loss_fwd/bwd`); MetricLoss (:81-137, not
used by any of the four approaches) is kept in plain torch."""
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
        if x.is_cuda and grad_output.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
            from isx import ops
            return ops.l2norm_rows_bwd(x, grad_output, ctx.eps), None
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


class _RowDeferredLinearFn(Function):
    """y = x W^T + b whose backward can hand the (x, dy) rows to the training step instead of forming dW itself (isx/dp.py RowSink)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from isx import dp
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        sink = dp.current_sink()
        ctx.shard = sink.shard_for(weight) if sink is not None and x.dim() == 2 else None
        if ctx.shard is not None:
            # the layer is sharded by output features across the ranks this step (isx/shard_head.py): rows and column slices are exchanged inside
            y, ctx.shard_ctx = ctx.shard.forward(x, sink.leaf_ids)
            return y
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        from isx import dp
        x, weight = ctx.saved_tensors
        if ctx.shard is not None:
            dy = dy.contiguous()
            dx = ctx.shard.backward(ctx.shard_ctx, dy)      # also records this pass's rows for the shard's weight update
            return dx, None, (dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None)
        x2, dy2 = x.reshape(-1, x.size(-1)), dy.reshape(-1, dy.size(-1)).contiguous()      # any leading dimensions, as nn.Linear
        dx = dy2.mm(weight).view_as(x) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            sink = dp.current_sink()
            if sink is not None and sink.accepts(weight):
                sink.add(weight, x2, dy2)                  # dW = dY^T X is formed ONCE per optimizer step, over every micro-batch's (and rank's) rows
            else:
                dw = dp.weight_gradient_from_rows(dy2, x2)
        db = dy2.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


class RowDeferredLinear(nn.Linear):
    """nn.Linear (same parameters, same state-dict keys) for the descriptor head `Linear(100352 -> 2048)` (reference
    model/siamese.py:104-114): ONE 822 MB weight.  Inside a training step (utils/train_general._Stepper) its weight gradient is not
    accumulated micro-batch by micro-batch (eight read-modify-write passes over 822 MB) nor all-reduced (822 MB twice per rank): the
    (x, dy) rows are collected -- 9.8 MB per micro-batch -- all-gathered across the ranks, and dW = dY^T X is formed once per step.
    Outside a step (no RowSink active) it behaves exactly like nn.Linear."""

    def forward(self, x):
        if torch.is_grad_enabled() and (self.weight.requires_grad or x.requires_grad):
            return _RowDeferredLinearFn.apply(x, self.weight, self.bias)
        return torch.nn.functional.linear(x, self.weight, self.bias)


# ---------------------------------------------------------------------------------------- losses
def _triplet_rows(anchor, pos, neg, margin, normalized):
    """Per-row clamped loss (reference custom_modules.py:153-167)."""
    if anchor.is_cuda:
        from isx import ops
        return ops.triplet_loss_rows(anchor, pos, neg, margin, normalized)
    if normalized:
        l = (anchor * neg).sum(1) - (anchor * pos).sum(1) + margin
    else:
        l = ((anchor - pos).pow(2).sum(1) - (anchor - neg).pow(2).sum(1) + 2 * margin) / 2
    return l.clamp(min=0)


class _Triplet(Function):
    @staticmethod
    def forward(ctx, anchor, pos, neg, margin, size_average, normalized):
        a, p, n = anchor.detach().float(), pos.detach().float(), neg.detach().float()
        rows = _triplet_rows(a, p, n, margin, normalized)
        ctx.save_for_backward(a, p, n, rows)
        ctx.cfg = (size_average, normalized)
        loss = rows.sum().view(1)
        return loss / a.size(0) if size_average else loss

    @staticmethod
    def backward(ctx, grad_output):
        a, p, n, rows = ctx.saved_tensors
        size_average, normalized = ctx.cfg
        if a.is_cuda and grad_output.is_cuda:
            # grad_output stays on the device (a host read-back here is one synchronisation per micro-batch of the training step)
            from isx import ops
            ga, gp, gn = ops.triplet_loss_grads(a, p, n, rows, 1.0 / a.size(0) if size_average else 1.0, normalized,
                                                scale_dev=grad_output.detach().float())
            return ga, gp, gn, None, None, None
        scale = float(grad_output.reshape(-1)[0]) / (a.size(0) if size_average else 1)
        if a.is_cuda:
            from isx import ops
            ga, gp, gn = ops.triplet_loss_grads(a, p, n, rows, scale, normalized)
        else:
            on = (rows > 0).float().view(-1, 1) * scale
            ga = (n - p) * on
            gp = (-a if normalized else p - a) * on
            gn = (a if normalized else a - n) * on
        return ga, gp, gn, None, None, None


class TripletLossFun(object):
    """sum_i max(0, a_i.n_i - a_i.p_i + margin) for unit vectors (normalized=True), else the squared
    distance form; `size_average` divides by the batch size."""

    def __init__(self, margin, size_average=True, normalized=True):
        self.margin, self.size_average, self.normalized = margin, size_average, normalized

    def __call__(self, anchor, pos, neg):
        return _Triplet.apply(anchor, pos, neg, self.margin, self.size_average, self.normalized)

    forward = __call__


class TripletLoss(nn.Module):
    def __init__(self, margin, size_average=True, normalized=True):
        super().__init__()
        self.margin, self.size_average, self.normalized = margin, size_average, normalized

    def forward(self, anchor, pos, neg):
        return _Triplet.apply(anchor, pos, neg, self.margin, self.size_average, self.normalized)


class MetricLossFun(object):
    """Chopra et al. contrastive loss with Q = 2 on the L1 energy E = |x1 - x2|_1:
    (1+y)/2 * E^2 + (1-y) * 2 * exp(-2.77 E / 2), y = +1 (same) / -1 (different).  Plain torch autograd."""

    def __init__(self, size_average=True):
        self.size_average = size_average

    def __call__(self, input1, input2, y):
        energy = (input1 - input2).abs().sum(1)
        loss = energy * energy * (1 + y) / 2 + torch.exp(-2.77 * energy / 2) * (1 - y) * 2
        loss = loss.sum().view(1)
        return loss / y.size(0) if self.size_average else loss

    forward = __call__


class MetricLoss(nn.Module):
    def __init__(self, size_average=True):
        super().__init__()
        self.size_average = size_average

    def forward(self, input1, input2, target):
        return MetricLossFun(self.size_average)(input1, input2, target)
