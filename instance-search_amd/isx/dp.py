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

Collective discipline: every rank issues exactly one all-reduce per bucket per step, ALWAYS in bucket order
0, 1, 2, ... -- whether a bucket was launched early from a hook (armed ranks) or from finish() (a rank whose
slice of the mini-batch was empty never arms; parameters no backward touched never fire a hook).  RCCL matches
collectives by issue order, so the order must not depend on what a rank happened to compute.
"""
import torch
import torch.distributed as dist

# bytes this rank sent + received in the exchanges of the most recent optimizer step, by leg (tools/bench_train.py prints them)
STATS = {}


def broadcast_module_state(module, src=0, group=None):
    """Make every replica start from rank `src`'s parameters AND buffers (BatchNorm statistics, counters).
    Without this each process keeps its own random initialisation of the layers that are not loaded from a file
    (the descriptor head) and the summed gradient is applied to different weights."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in module.state_dict().values():          # state_dict tensors share storage with the module
            dist.broadcast(t, src=src, group=group)


def batch_norm_buffers(module):
    """Floating-point buffers (running_mean / running_var) of every BatchNorm that is in TRAINING mode under `module`."""
    out = []
    for m in module.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training and m.track_running_stats:
            out += [b for b in (m.running_mean, m.running_var) if b is not None]
    return out


def average_buffers(buffers, group=None):
    """Replace each buffer by its mean over the ranks (one flat all-reduce).  Data-parallel training with BatchNorm in training mode
    (P.train_bn): every rank updates its running statistics from its own slice of the mini-batch; without this the replicas drift apart,
    mining / evaluation differ per rank and the checkpoint written by rank 0 carries rank 0's statistics only."""
    if not buffers or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        flat = torch.cat([b.reshape(-1).float() for b in buffers])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= dist.get_world_size(group)
        off = 0
        for b in buffers:
            n = b.numel()
            b.copy_(flat[off:off + n].view_as(b))
            off += n


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
            self.slices[p] = (off, off + n)
            off += n
        self._attach_all()
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
        self.ranges = [(min(self.slices[q][0] for q in ps), max(self.slices[q][1] for q in ps)) for ps in self.buckets]
        self.pending = [0] * len(self.buckets)
        self.next_bucket = 0              # buckets [0, next_bucket) of this step have been issued
        self.handles = []
        self.armed = False
        if self.world > 1:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    # -- the gradients must stay views of the flat buffer ------------------------------------------------------
    def _view(self, p):
        lo, hi = self.slices[p]
        return self.flat[lo:hi].view_as(p)

    def _is_view(self, p):
        lo, _ = self.slices[p]
        return p.grad is not None and p.grad.data_ptr() == self.flat.data_ptr() + 4 * lo and p.grad.dtype == torch.float32

    def _reattach(self, p):
        """`optimizer.zero_grad(set_to_none=True)` or `p.grad = None` elsewhere detaches a gradient from the flat buffer;
        autograd then accumulates into a fresh tensor the exchange would never see.  Fold it back in."""
        if self._is_view(p):
            return
        v = self._view(p)
        if p.grad is not None:
            v.copy_(p.grad)                 # this step's gradient so far lives in the stray tensor (whoever detached the view discarded what it held)
        else:
            v.zero_()                       # no gradient this step: the slice still holds the PREVIOUS step's summed gradient -- it must not be exchanged and applied again
        p.grad = v

    def _attach_all(self):
        for p in self.params:
            self._reattach(p)

    def zero_grad(self):
        self._attach_all()
        self.flat.zero_()

    # -- exchange ---------------------------------------------------------------------------------------------
    def arm(self):
        """Call before the backward of the LAST micro-batch of a step: buckets are exchanged as they fill."""
        self.pending = [len(ps) for ps in self.buckets]
        self.armed = self.world > 1

    def _issue_ready(self, upto_complete_only):
        """Issue the all-reduce of every not-yet-issued bucket, in bucket order; with `upto_complete_only` stop at the
        first bucket that still waits for gradients."""
        while self.next_bucket < len(self.buckets):
            b = self.next_bucket
            if upto_complete_only and self.pending[b] > 0:
                break
            lo, hi = self.ranges[b]
            self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.next_bucket += 1

    def _hook(self, p):
        if not self.armed:
            return
        self._reattach(p)
        b = self.bucket_of[p]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._issue_ready(True)

    def finish(self):
        """Issue whatever the hooks did not (unarmed rank, unused parameters), wait for the exchange."""
        if self.world > 1:
            self._attach_all()
            self._issue_ready(False)
            for h in self.handles:
                h.wait()
        self.armed = False
        self.handles = []
        self.next_bucket = 0
