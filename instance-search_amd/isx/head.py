"""The descriptor head of DescriptorNet (reference model/siamese.py:104-123: NormalizeL2 -> Shift -> Linear(100352 -> D) -> NormalizeL2)
for ALL local micro-batches of a training step at once, forward and backward by hand over libisx.

torch (and the reference, utils/train_general.py:51-74) run the head once per micro-batch of 8 triplets: ~25 small launches and two passes
over the 822 MB weight each -- eight times per step: the head was the host-bound AND the most HBM-hungry part of the step (16 weight
passes, 13 GB).  Here the rows of every local micro-batch go through ONE pass:

  forward    flatten -> isx_l2norm_shift_rows -> isx_head_linear_fwd (split-K GEMM whose per-row result does not depend on the row count)
             -> isx_l2norm_rows: descriptors (M, D).  The loss stays the training script's own callback, evaluated per micro-batch on
             its rows of the descriptors (a tiny autograd graph: the triplet kernels), which yields d(loss)/d(descriptors).
  backward   isx_l2norm_rows_bwd -> rows (x, dy) to the step's RowSink (the 822 MB weight gradient is formed once per step, isx/dp.py)
             -> isx_colsum_leaves (bias gradient PER micro-batch) -> isx_head_linear_dgrad -> isx_colsum_leaves (Shift gradient per
             micro-batch) -> isx_l2norm_rows_bwd: gradient wrt the trunk output.

Every kernel computes a row exactly as it would alone and the per-micro-batch sums run over that micro-batch's rows in order, so the
gradients of a micro-batch are the same bits whether 1 or 8 micro-batches share the pass -- the property the canonical gradient tree
(isx/dp.py) needs to keep the update independent of the number of ranks.
"""
import torch

from . import _lib, ops
from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class HeadEngine(object):
    def __init__(self, net):
        self.net = net
        self.shift = net.feature_reduc1[1]
        self.lin = net.feature_reduc1[2]
        self._ws = None

    @staticmethod
    def applicable(net):
        from model.custom_modules import NormalizeL2, RowDeferredLinear, Shift
        h = getattr(net, "feature_reduc1", None)
        if h is None or len(h) != 3 or not (isinstance(h[0], NormalizeL2) and isinstance(h[1], Shift) and isinstance(h[2], RowDeferredLinear)):
            return False
        if not isinstance(getattr(net, "feature_reduc2", None), NormalizeL2):
            return False
        lin = h[2]
        return (lin.weight.is_cuda and lin.weight.dtype == torch.float32 and lin.weight.is_contiguous() and lin.in_features % 64 == 0
                and lin.out_features % 64 == 0 and h[1].param.numel() == lin.in_features)

    def forward(self, f_all, shard=None, leaf_ids=None):
        """f_all: (M, C, h, w) trunk output of all local rows (no graph).  Returns (descriptors (M, D), context for backward).
        shard / leaf_ids: the Linear is sharded by output features across the ranks this step (isx/shard_head.HeadShard) -- the rows of every
        rank go through this rank's slice of the weight, the column slices are exchanged."""
        M = f_all.size(0)
        lin = self.lin
        x0 = f_all.reshape(M, -1)                                   # logical (C, h, w) order whatever the memory format
        if not x0.is_contiguous():
            x0 = x0.contiguous()
        x1 = ops.l2norm_shift_rows(x0, self.shift.param.detach())
        b = lin.bias
        sctx = None
        if shard is not None:
            y, sctx = shard.forward(x1, leaf_ids)
            y = y.contiguous()
        else:
            y = ops.head_linear(x1, lin.weight.detach(), b.detach() if b is not None else None)      # rows as stored: no transposed copy
        d = ops.l2norm_rows(y)
        return d, (x0, x1, y, tuple(f_all.shape), shard, sctx)

    def backward(self, ctx, dd, leaves, sink, flat_all, slices):
        """dd: (M, D) gradient wrt the descriptors; `leaves` consecutive micro-batches of equal row count.  The small parameters' gradients
        go to row l of flat_all (per-leaf flat gradient buffers) at the parameters' slices, the Linear's (x, dy) rows to `sink`.
        Returns the gradient wrt the trunk output, shaped (M, C, h, w)."""
        x0, x1, y, shape, shard, sctx = ctx
        M, K = x1.shape
        lin = self.lin
        N = lin.out_features
        R = M // leaves
        if R * leaves != M:
            raise _lib.IsxError("head engine: %d rows are not %d equal micro-batches" % (M, leaves))
        dy = ops.l2norm_rows_bwd(y, dd.contiguous())
        if lin.weight.requires_grad and shard is None:
            if sink is None or not sink.accepts(lin.weight):
                raise _lib.IsxError("head engine: the Linear weight needs the training step's RowSink")
            sink.add(lin.weight, x1, dy)
        if lin.bias is not None and lin.bias.requires_grad:
            gb = torch.empty((leaves, N), dtype=torch.float32, device=dy.device)
            check(lib().isx_colsum_leaves(dy.data_ptr(), leaves, R, N, gb.data_ptr(), _stream()), "isx_colsum_leaves")
            lo, hi = slices[lin.bias]
            flat_all[:, lo:hi].copy_(gb)
        if shard is not None:
            dx1 = shard.backward(sctx, dy).contiguous()        # every rank's chains of its feature groups, this rank's rows summed in group order
        else:
            Mp = (M + 63) // 64 * 64
            if Mp == M:
                dyT = dy.t().contiguous()
            else:
                dyT = dy.new_zeros((N, Mp))
                dyT[:, :M] = dy.t()
            dx1 = torch.empty((Mp, K), dtype=torch.float32, device=dy.device)
            check(lib().isx_head_linear_dgrad(dyT.data_ptr(), Mp, N, lin.weight.data_ptr(), K, dx1.data_ptr(), _stream()), "isx_head_linear_dgrad")
            dx1 = dx1[:M]
        if self.shift.param.requires_grad:
            gs = torch.empty((leaves, K), dtype=torch.float32, device=dy.device)
            check(lib().isx_colsum_leaves(dx1.data_ptr(), leaves, R, K, gs.data_ptr(), _stream()), "isx_colsum_leaves")
            lo, hi = slices[self.shift.param]
            flat_all[:, lo:hi].copy_(gs)
        dx0 = ops.l2norm_rows_bwd(x0, dx1)
        return dx0.view(shape)
