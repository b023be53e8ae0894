"""Gradient accumulation and data-parallel exchange for the siamese training step (BASELINE config 4).

One process per GPU.  The reference accumulates the gradients of 8 micro-batches of 8 triplets before every SGD step
(train_batch_size 64 / train_micro_batch 8, utils/train_general.py:51-74, loss summed not averaged) on ONE device.  Here the
micro-batches of a mini-batch are the LEAVES of a fixed binary tree:

    leaves   = the micro-batches of the mini-batch, in order (leaf i = triplets [i*mb, (i+1)*mb))
    node     = left subtree + right subtree, split at lo + (hi - lo) // 2            (`tree_split`)
    rank r   = the depth-log2(P) subtree with index r                                (`rank_leaves`)

A rank sums its own leaves in tree order (`tree_sum`), `TreeExchange` finishes the upper log2(P) levels across the ranks in the
SAME order, so the summed gradient -- and with it every weight after every step -- is bit-identical for P = 1, 2, 4, 8
(tests/test_training.py: 2-rank gloo run == single process, torch.equal on the whole state dict).  An all-reduce cannot give that:
its summation order is RCCL's, and a rank's partial sum is a different rounding of the same micro-batches.

Two kinds of gradient travel:

  * everything but the descriptor head's weight -- ResNet-50 layer4 (15 M parameters, 60 MB) + Shift + biases -- lives in ONE flat fp32
    buffer (`FlatGrads`: each `param.grad` is a view into it).  TreeExchange moves it as all-to-all (slice j of every rank -> rank j),
    a tree-ordered sum of the P received pieces, and an all-gather of the reduced slices: the bytes of a ring all-reduce
    (2 (P-1)/P x 60 MB per rank), deterministic order.
  * the head `Linear(100352 -> 2048)` is ONE 822 MB weight (reference model/siamese.py:104-114).  Its gradient is never exchanged:
    the head's `RowDeferredLinear` hands the (x, dy) ROWS of every micro-batch to the step's `RowSink`; the ranks all-gather the rows
    (24 x (100352 + 2048) x 4 B = 9.8 MB per micro-batch, 79 MB per mini-batch instead of 822 MB twice) and every rank forms
    dW = dY^T X over the rows of the whole mini-batch in micro-batch order -- one GEMM, the same rows in the same order at any P, so
    the same bits.  On one GPU the same deferral replaces eight read-modify-write passes over the 822 MB gradient by one write.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): at P = 8 a rank receives 7 x 9.8 MB of rows + 2 x 52 MB of flat slices per step,
~0.4 ms, against ~10 ms for the 822 MB ring all-reduce this replaces.

`GradAllReducer` (sequential accumulation + bucketed all-reduce(SUM) overlapped with the last backward) stays for world sizes that are not
a power of two and as the A/B (`P.train_grad_exchange = "allreduce"`); its buckets are BYTE RANGES of the flat buffer, not parameter
groups, so one large parameter no longer makes one large bucket.  Collective discipline there: every rank issues exactly one all-reduce
per bucket per step, ALWAYS in bucket order 0, 1, 2, ... -- RCCL matches collectives by issue order.
"""
import torch
import torch.distributed as dist

# bytes this rank sent / received in the exchanges of the most recent optimizer step, by leg (tools/bench_train.py prints them)
STATS = {}


def _world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def broadcast_module_state(module, src=0, group=None):
    """Make every replica start from rank `src`'s parameters AND buffers (BatchNorm statistics, counters).
    Without this each process keeps its own random initialisation of the layers that are not loaded from a file
    (the descriptor head) and the summed gradient is applied to different weights."""
    if _world(group) == 1:
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
    if not buffers or _world(group) == 1:
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


# ---- the canonical tree ---------------------------------------------------------------------------------------------------------
def tree_split(lo, hi):
    return lo + (hi - lo) // 2


def is_power_of_two(n):
    return n >= 1 and (n & (n - 1)) == 0


def rank_leaves(n_leaves, world, rank):
    """[lo, hi): the leaves of the subtree that belongs to `rank` (world a power of two): walk log2(world) levels down the tree, halving
    the rank range and the leaf range together.  With fewer leaves than ranks some ranks own nothing (lo == hi)."""
    if not is_power_of_two(world):
        raise ValueError("the canonical gradient tree needs a power-of-two world size, got %d" % world)
    lo, hi, r0, r1 = 0, n_leaves, 0, world
    while r1 - r0 > 1:
        rm, mid = (r0 + r1) // 2, tree_split(lo, hi)
        if rank < rm:
            hi, r1 = mid, rm
        else:
            lo, r0 = mid, rm
    return lo, hi


def tree_sum(lo, hi, leaf):
    """Sum of leaf(lo) ... leaf(hi - 1) in tree order; `leaf(i)` returns a tensor this function may add into (leaves are evaluated in
    increasing i).  None for an empty range."""
    if hi <= lo:
        return None
    if hi - lo == 1:
        return leaf(lo)
    mid = tree_split(lo, hi)
    a = tree_sum(lo, mid, leaf)
    b = tree_sum(mid, hi, leaf)
    a.add_(b)
    return a


class FlatGrads(object):
    """All given trainable parameters' gradients as views of ONE flat fp32 buffer (parameter order)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self.slices = {}
        for p in self.params:
            n = p.numel()
            self.slices[p] = (off, off + n)
            off += n
        self.attach_all()

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

    def attach_all(self):
        for p in self.params:
            self._reattach(p)

    def zero_grad(self):
        self.attach_all()
        self.flat.zero_()

    def take(self):
        """The gradient accumulated since the last zero (a copy), leaving the buffer zeroed for the next leaf."""
        self.attach_all()
        out = self.flat.clone()
        self.flat.zero_()
        return out

    def put(self, total):
        self.attach_all()
        if total is None:
            self.flat.zero_()
        else:
            self.flat.copy_(total)


class TreeExchange(object):
    """Sum of every rank's flat buffer in the canonical tree order, result on every rank.  all-to-all (slice j -> rank j) + tree-ordered
    sum of the P pieces + all-gather of the reduced slices; point-to-point friendly (each pair of GPUs exchanges 2 x n/P floats)."""

    def __init__(self, group=None):
        self.group = group
        self.world = _world(group)
        self.rank = _rank(group)
        if not is_power_of_two(self.world):
            raise ValueError("TreeExchange needs a power-of-two world size")
        self._pad = None

    def allreduce_(self, flat):
        P = self.world
        if P == 1 or flat.numel() == 0:
            return flat
        n = flat.numel()
        per = -(-n // P)
        if self._pad is None or self._pad.numel() != per * P or self._pad.device != flat.device:
            self._pad = torch.zeros(per * P, dtype=flat.dtype, device=flat.device)
        pad = self._pad
        pad[:n].copy_(flat)
        moved = 4 * per * (P - 1)
        STATS["flat_all_to_all_bytes_sent"] = moved
        STATS["flat_all_gather_bytes_received"] = moved
        STATS["flat_gradient_bytes"] = 4 * n
        if flat.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo has no all-to-all on device tensors: stage through the host (two ranks on ONE GPU in tests/test_training.py; RCCL never gets here)
            host = pad.cpu()
            recv = torch.empty_like(host)
            dist.all_to_all_single(recv, host, group=self.group)
            pieces = recv.view(P, per)
            red = tree_sum(0, P, lambda j: pieces[j])                     # same order, same fp32 adds as on the device
            dist.all_gather_into_tensor(host, red.contiguous(), group=self.group)
            pad.copy_(host)
            flat.copy_(pad[:n])
            return flat
        recv = torch.empty_like(pad)
        dist.all_to_all_single(recv, pad, group=self.group)               # recv.view(P, per)[j] = rank j's slice number self.rank
        pieces = recv.view(P, per)
        red = tree_sum(0, P, lambda j: pieces[j])                         # in place into the received pieces: rank order = subtree order
        dist.all_gather_into_tensor(pad, red.contiguous(), group=self.group)
        flat.copy_(pad[:n])
        return flat


# ---- row-deferred weight gradients ---------------------------------------------------------------------------------------------------
_SINK = None


def current_sink():
    return _SINK


class RowSink(object):
    """Collects the (x, dy) rows `RowDeferredLinear` produces during the backward passes of ONE optimizer step, in the order they are
    produced (micro-batch order), and turns them into the weight gradients at the end of the step."""

    def __init__(self, weights, group=None, optimizer=None, shards=None):
        """optimizer: the step's torch.optim.SGD, or None.  With it, a deferred weight that libisx can handle is UPDATED by the kernel that forms
        its gradient (isx_head_sgd_step: no gradient tensor, `weight.grad` stays None and the optimizer's own step skips the weight).
        shards: {id(weight): isx.shard_head.HeadShard} -- weights sharded by output features across the ranks this step (their rows never come
        here: the shard exchanges them itself); leaf_ids: the global micro-batch indices of the rows about to go through a sharded head."""
        self.shards = dict(shards or {})
        self.leaf_ids = []
        self.weights = list(weights)
        self.ids = dict((id(w), k) for k, w in enumerate(self.weights))
        self.group = group
        self.optimizer = optimizer
        self.x = [[] for _ in self.weights]
        self.dy = [[] for _ in self.weights]

    def accepts(self, weight):
        return id(weight) in self.ids

    def shard_for(self, weight):
        return self.shards.get(id(weight))

    def add(self, weight, x, dy):
        k = self.ids[id(weight)]
        self.x[k].append(x.detach())
        self.dy[k].append(dy.detach())

    def __enter__(self):
        global _SINK
        self._prev, _SINK = _SINK, self
        return self

    def __exit__(self, *exc):
        global _SINK
        _SINK = self._prev
        return False

    @staticmethod
    def _gather_rows(t, group):
        """Rows of every rank, rank order (= micro-batch order).  Row counts may differ per rank (uneven leaves)."""
        P = _world(group)
        if P == 1:
            return t
        cnt = torch.tensor([t.size(0)], dtype=torch.int64, device=t.device)
        cnts = torch.empty(P, dtype=torch.int64, device=t.device)
        dist.all_gather_into_tensor(cnts, cnt, group=group)
        cnts = cnts.tolist()
        mx = max(cnts)
        if mx == 0:
            return t
        if t.size(0) < mx:
            t = torch.cat([t, t.new_zeros((mx - t.size(0),) + tuple(t.shape[1:]))], 0)
        out = t.new_empty((P * mx,) + tuple(t.shape[1:]))
        dist.all_gather_into_tensor(out, t.contiguous(), group=group)
        STATS["rows_all_gather_bytes_received"] = STATS.get("rows_all_gather_bytes_received", 0) + out[0].numel() * 4 * mx * (P - 1)
        if all(c == mx for c in cnts):
            return out
        return torch.cat([out[j * mx:j * mx + c] for j, c in enumerate(cnts)], 0)

    def finish(self):
        """dW = dY^T X over the rows of the whole mini-batch (every rank's, rank order) -> weight.grad, for every deferred weight."""
        STATS.pop("rows_all_gather_bytes_received", None)
        STATS["row_deferred_weight_bytes"] = sum(4 * w.numel() for w in self.weights)      # what an all-reduce of these gradients would move (x 2 (P-1)/P)
        for sh in self.shards.values():
            sh.finish(self.optimizer)
        with torch.no_grad():
            for k, w in enumerate(self.weights):
                if id(w) in self.shards:
                    continue
                if self.x[k]:
                    x, dy = torch.cat(self.x[k], 0), torch.cat(self.dy[k], 0)
                else:
                    x, dy = w.new_zeros((0, w.size(1))), w.new_zeros((0, w.size(0)))
                x, dy = self._gather_rows(x, self.group), self._gather_rows(dy, self.group)
                self.x[k], self.dy[k] = [], []
                if self.optimizer is not None and fused_sgd_from_rows(self.optimizer, w, dy, x):
                    continue
                g = weight_gradient_from_rows(dy, x)
                if w.grad is not None and w.grad.shape == g.shape and w.grad.dtype == g.dtype:
                    w.grad.copy_(g)
                else:
                    w.grad = g


def fused_sgd_from_rows(optimizer, w, dy, x):
    """Weight gradient dy^T x over the rows of the mini-batch + torch.optim.SGD's update of `w` in ONE libisx kernel (isx_head_sgd_step): the
    gradient is never written.  The momentum buffer is the optimizer's own state entry (created on the first step, as torch does), so a
    state_dict / a re-created optimizer (annealing) behaves as with torch's step.  Returns False -- nothing done -- when the case is not the
    kernel's (CPU tensors, widths that are not multiples of 128, an optimizer that is not a plain SGD, maximize / differentiable)."""
    if not (type(optimizer) is torch.optim.SGD and w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 2
            and w.size(0) % 64 == 0 and w.size(1) % 128 == 0 and 128 * w.size(1) * 4 < 2 ** 31 and x.dtype == torch.float32 and dy.dtype == torch.float32):
        return False
    group = next((g for g in optimizer.param_groups if any(p is w for p in g['params'])), None)
    if group is None or group.get('maximize') or group.get('differentiable'):
        return False
    from ._lib import check, lib
    mom, first, buf = float(group['momentum']), 0, None
    if mom != 0.0:
        state = optimizer.state[w]
        buf = state.get('momentum_buffer')
        if buf is None:
            buf = state['momentum_buffer'] = torch.empty_like(w)       # written in full by the first step (buf = g)
            first = 1
    x, dy = x.contiguous(), dy.contiguous()
    lr = group['lr']
    check(lib().isx_head_sgd_step(dy.data_ptr(), x.data_ptr(), x.size(0), w.size(0), w.size(1), w.data_ptr(), buf.data_ptr() if buf is not None else None, first,
                                  float(lr), mom, float(group['dampening']), float(group['weight_decay']), 1 if group['nesterov'] else 0,
                                  torch.cuda.current_stream().cuda_stream), "isx_head_sgd_step")
    w.grad = None                    # the optimizer's step skips a parameter without a gradient
    bump_version(w)
    return True


def bump_version(t):
    """A libisx kernel (or a write through .data) changed `t` behind autograd's back: move its version counter as an in-place op would, for the
    caches keyed on it (model/nn_utils._derived, RegionDescriptorNet._hwc_head)."""
    try:
        torch._C._autograd._unsafe_set_version_counter([t], [t._version + 1])
    except (AttributeError, TypeError):
        pass


def weight_gradient_from_rows(dy, x):
    """(R, out), (R, in) -> (out, in) = dy^T x.  On the GPU: libisx's TN GEMM over the rows (isx_conv_wgrad_nhwc with the rows as "pixels":
    both operands are K-major as stored, every output is ONE fp32 fma chain over the rows in row order -- deterministic by construction);
    otherwise (CPU, widths that are not multiples of 64) one torch GEMM of a fixed shape."""
    if x.size(0) == 0:
        return x.new_zeros((dy.size(1), x.size(1)))
    R, n_out, n_in = x.size(0), dy.size(1), x.size(1)
    if x.is_cuda and x.dtype == torch.float32 and dy.dtype == torch.float32 and n_out % 64 == 0 and n_in % 64 == 0:
        from ._lib import check, lib
        L = lib()
        if L.isx_conv_wgrad_splits(R, n_in, n_out, 1) == 1:
            x, dy = x.contiguous(), dy.contiguous()
            g = torch.empty((n_out, n_in), dtype=torch.float32, device=x.device)
            check(L.isx_conv_wgrad_nhwc(dy.data_ptr(), x.data_ptr(), R, 1, 1, 1, n_in, n_out, 1, 1, g.data_ptr(), None,
                                        torch.cuda.current_stream().cuda_stream), "isx_conv_wgrad_nhwc")
            return g
    return dy.t().mm(x)


# ---- all-reduce path (A/B; world sizes that are not a power of two) -----------------------------------------------------------------
class GradAllReducer(FlatGrads):
    """Sequentially accumulated gradients in the flat buffer + bucketed all-reduce(SUM).  The flat buffer is laid out in REVERSE
    parameter order (the order in which backward produces gradients) and cut into buckets of `bucket_mb` BY BYTE RANGE; a bucket's
    all-reduce is launched asynchronously from the autograd hook that completes it on the FINAL micro-batch, so the exchange of early
    buckets overlaps the rest of the backward pass."""

    def __init__(self, params, group=None, bucket_mb=64):
        self.group = group
        self.world = _world(group)
        super().__init__(list(reversed([p for p in params if p.requires_grad])))
        self.params = list(reversed(self.params))                # public order = parameter order (the views do not move)
        n = self.flat.numel()
        cap = max(1, int(bucket_mb * (1 << 20) // 4))
        self.ranges = [(lo, min(lo + cap, n)) for lo in range(0, n, cap)]
        # parameters overlapping each byte range; the bucket is complete when all of them have their gradient
        self.members = [[p for p in self.params if self.slices[p][0] < hi and self.slices[p][1] > lo] for lo, hi in self.ranges]
        self.buckets = self.members
        self.buckets_of = {}
        for b, ps in enumerate(self.members):
            for p in ps:
                self.buckets_of.setdefault(p, []).append(b)
        self.pending = [0] * len(self.ranges)
        self.next_bucket = 0              # buckets [0, next_bucket) of this step have been issued
        self.handles = []
        self.armed = False
        if self.world > 1:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)

    _attach_all = FlatGrads.attach_all

    def arm(self):
        """Call before the backward of the LAST micro-batch of a step: buckets are exchanged as they fill."""
        self.pending = [len(ps) for ps in self.members]
        self.armed = self.world > 1

    def _issue_ready(self, upto_complete_only):
        """Issue the all-reduce of every not-yet-issued bucket, in bucket order; with `upto_complete_only` stop at the
        first bucket that still waits for gradients."""
        while self.next_bucket < len(self.ranges):
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
        for b in self.buckets_of[p]:
            self.pending[b] -= 1
        self._issue_ready(True)

    def finish(self):
        """Issue whatever the hooks did not (unarmed rank, unused parameters), wait for the exchange."""
        if self.world > 1:
            self.attach_all()
            self._issue_ready(False)
            for h in self.handles:
                h.wait()
            STATS["flat_all_reduce_bytes"] = 4 * self.flat.numel()
        self.armed = False
        self.handles = []
        self.next_bucket = 0
