"""Data-parallel gradient exchange for the siamese training step (BASELINE config 4).

One process per GPU.  The reference accumulates the gradients of 8 micro-batches of 8 triplets before
every SGD step (train_batch_size 64 / train_micro_batch 8, utils/train_general.py:51-74, loss summed
not averaged); with P ranks each rank takes 1/P of the mini-batch's triplets and the summed gradient
is recovered with an all-reduce(SUM) -- the same update, P times sooner.

All trainable gradients live in ONE flat fp32 buffer (each `param.grad` is a view into it), cut into
buckets of `bucket_mb`; a bucket's all-reduce is launched asynchronously from the autograd hook of the
last of its parameters on the FINAL micro-batch, so the exchange of early buckets overlaps the rest of
the backward pass.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): ~0.9 GB of ResNet-50 +
Linear(100352, 2048) gradients is ~10 ms on one ring, so few, large buckets are used (default 256 MB).
"""
import torch
import torch.distributed as dist


class GradAllReducer(object):
    def __init__(self, params, group=None, bucket_mb=256):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        # gradients become views of the flat buffer; buckets follow REVERSE parameter order (the order
        # in which backward produces them)
        off = 0
        self.slices = {}
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.slices[p] = (off, off + n)
            off += n
        cap = max(1, int(bucket_mb * (1 << 20) // 4))
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel()
            if size >= cap:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {p: b for b, ps in enumerate(self.buckets) for p in ps}
        self.pending = [0] * len(self.buckets)
        self.handles = []
        self.armed = False
        if self.world > 1:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    def zero_grad(self):
        self.flat.zero_()

    def arm(self):
        """Call before the backward of the LAST micro-batch of a step: buckets are exchanged as they fill."""
        self.pending = [len(ps) for ps in self.buckets]
        self.handles = []
        self.armed = self.world > 1

    def _hook(self, p):
        if not self.armed:
            return
        b = self.bucket_of[p]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            lo = min(self.slices[q][0] for q in self.buckets[b])
            hi = max(self.slices[q][1] for q in self.buckets[b])
            self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for the exchange (and exchange whatever the hooks did not cover, e.g. unused parameters)."""
        if self.world > 1:
            if self.armed:
                for b, left in enumerate(self.pending):
                    if left > 0:
                        lo = min(self.slices[q][0] for q in self.buckets[b])
                        hi = max(self.slices[q][1] for q in self.buckets[b])
                        self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                self.handles.append(dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            for h in self.handles:
                h.wait()
        self.armed = False
        self.handles = []
