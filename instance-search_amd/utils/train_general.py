"""Batch fold driver and the generic training loop (reference utils/train_general.py): fold_batches
(:27-38), anneal (:41-48), micro/mini-batch steps with gradient accumulation (:51-74), train_gen
(:77-105), output_stats (:12-23).  Added: data-parallel execution -- with torch.distributed
initialised every rank takes the micro-batches of its subtree of each mini-batch and the gradients are
summed in a fixed tree order across the ranks (isx/dp.py), bit-identical to the single-process run."""
import random

import torch
import torch.distributed as dist
import torch.optim as optim

from model.nn_utils import set_net_train
from .general import log


def fold_batches(f, init, x, batch_size, cut_end=False, add_args={}):
    """Left fold of `f(acc, start_index, is_final, x[start:end], **add_args)` over consecutive
    batches of `x`.  batch_size <= 0: one call on the whole set.  cut_end drops a trailing
    partial batch (and then flags the last FULL batch as final)."""
    n = len(x)
    if batch_size <= 0:
        return f(init, 0, True, x, **add_args)
    acc = init
    for start in range(0, n, batch_size):
        end = min(start + batch_size, n)
        if cut_end and start + batch_size > n:
            continue
        is_final = (end > n - batch_size) if cut_end else (end == n)
        acc = f(acc, start, is_final, x[start:end], **add_args)
    return acc


def output_stats(train_type, P, test_print, test_net, net, testset_tuple, epoch, batch_count, is_final, loss, running_loss, score):
    """Running-loss line every P.train_loss_int mini-batches; evaluation every P.train_test_int (or at
    the end of the epoch when that is <= 0)."""
    every = P.train_loss_int
    running_loss = running_loss + loss              # `loss` may be a device scalar (_Stepper.step): summed where it lives, in float64
    if batch_count % every == every - 1:
        log(P, '[{0:d}, {1:5d}] loss: {2:.5f}'.format(epoch + 1, batch_count + 1, float(running_loss) / every))
        running_loss = 0.0
    t = P.train_test_int
    if (t > 0 and batch_count % t == t - 1) or (t <= 0 and is_final):
        score = test_print(train_type, P, net, testset_tuple, test_net, score, epoch + 1)
    return running_loss, score


def make_sgd(params, lr, momentum, weight_decay):
    """optim.SGD as the reference builds it (train/siamese_descriptor.py:190-191); on the GPU the FUSED implementation: one pass over
    (parameter, gradient, momentum buffer) per tensor list instead of torch's default chain of foreach kernels -- 882 MB of parameters make
    the default update 3.1 ms of a 39 ms step.  Same formula: d = g + wd p; buf = momentum buf + d; p -= lr buf."""
    params = list(params)
    fused = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
    return optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay, **({"fused": True} if fused else {}))


def anneal(net, optimizer, epoch, annealing_dict):
    """At the epochs listed in annealing_dict a NEW SGD is built with lr scaled by the given factor
    (momentum buffers start afresh, as in the reference)."""
    if epoch not in annealing_dict:
        return optimizer
    g = optimizer.state_dict()['param_groups'][0]
    return make_sgd((p for p in net.parameters() if p.requires_grad), g['lr'] * annealing_dict[epoch], g['momentum'], g['weight_decay'])


# Diagnostic: set to a dict to get synchronised wall-clock seconds per phase of the training step accumulated into it (tools/bench_train.py --phases);
# None (default): no synchronisation, no timing.
PHASES = None


class _phase(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if PHASES is not None:
            import time
            torch.cuda.synchronize()
            self.t = time.perf_counter()

    def __exit__(self, *exc):
        if PHASES is not None:
            import time
            torch.cuda.synchronize()
            PHASES[self.name] = PHASES.get(self.name, 0.0) + time.perf_counter() - self.t
        return False


def _dp():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


class _Stepper(object):
    """One optimizer step = the gradients of the micro-batches of a mini-batch (reference utils/train_general.py:51-74: micro_batch_gen /
    mini_batch_gen accumulate them one after the other), summed here in the canonical TREE order of isx/dp.py: a rank runs the
    micro-batches of its subtree, TreeExchange finishes the sum across the ranks in the same order, and the descriptor head's 822 MB
    weight gradient is formed once per step from the all-gathered (x, dy) rows (RowDeferredLinear / RowSink).  The update is therefore
    bit-identical for 1, 2, 4, 8 ranks.  `P.train_grad_exchange = "allreduce"` (and world sizes that are not a power of two): sequential
    accumulation over this rank's contiguous slice + bucketed all-reduce overlapped with the last backward (dp.GradAllReducer)."""

    def __init__(self, P, net, make_batch, make_loss):
        from isx import dp
        from model.custom_modules import RowDeferredLinear
        self.P, self.net, self.make_batch, self.make_loss = P, net, make_batch, make_loss
        self.rank, self.world = _dp()
        mode = getattr(P, 'train_grad_exchange', 'tree')
        if mode not in ('tree', 'allreduce'):
            raise ValueError("P.train_grad_exchange must be 'tree' or 'allreduce', got %r" % (mode,))
        if mode == 'tree' and not dp.is_power_of_two(self.world):
            log(P, 'world size {0} is not a power of two: gradient exchange falls back to all-reduce (not bit-identical to one process)'.format(self.world))
            mode = 'allreduce'
        self.mode = mode
        self.deferred = [m.weight for m in net.modules() if isinstance(m, RowDeferredLinear) and m.weight.requires_grad]
        rest = [p for p in net.parameters() if p.requires_grad and not any(p is w for w in self.deferred)]
        if mode == 'tree':
            self.flat = dp.FlatGrads(rest)
            self.exchange = dp.TreeExchange() if self.world > 1 else None
        else:
            self.flat = dp.GradAllReducer(rest)
            self.exchange = None
        # The descriptor head sharded by output features (isx/shard_head.py): rank r forms and applies the update of ITS rows of the 822 MB weight
        # only.  On the GPU the single process needs no shard object (its kernels already compute the canonical sums); on the CPU it does, so
        # that 1, 2, 4, 8 processes issue the same (micro-batch, feature group) GEMMs.
        self.shards = {}
        if mode == 'tree' and getattr(P, 'train_head_shard', True):
            from isx import shard_head
            for m in net.modules():
                if isinstance(m, RowDeferredLinear) and m.weight.requires_grad and shard_head.shardable(m.weight, self.world) \
                        and (self.world > 1 or not m.weight.is_cuda):
                    self.shards[id(m.weight)] = shard_head.HeadShard(m.weight, m.bias)

    def sync_head(self):
        """All ranks' rows of a sharded head to every rank (before anything reads the whole weight: embedding pass, evaluation, checkpoint)."""
        for sh in self.shards.values():
            sh.sync()

    # -- one micro-batch ---------------------------------------------------------------------------------------------------------
    def _forward_backward(self, triplets, offset, mini_size, batch_args, pre, arm=False):
        if pre is not None:
            # prefix features of the whole local slice were computed in one launch (frozen trunk prefix): this micro-batch takes its rows
            feats, targets_all = pre
            n = len(triplets)
            out = self.net.forward_features(*[f[offset:offset + n] for f in feats])
            targets = [t[offset:offset + n] if torch.is_tensor(t) and t.dim() > 0 and t.size(0) == feats[0].size(0) else t for t in targets_all]
            loss, loss2 = self.make_loss(out, targets)
        else:
            inputs, targets = self.make_batch(triplets, len(triplets), **batch_args)
            loss, loss2 = self.make_loss(self.net(*inputs), targets)
        P = self.P
        share = len(triplets) / float(mini_size)
        obj = loss * share if P.train_loss_avg else loss
        if loss2 is not None:
            obj = obj + P.train_loss2_alpha * (loss2 * share if P.train_loss2_avg else loss2)
        if arm:
            self.flat.arm()                          # all-reduce path: exchange buckets as this last backward fills them
        obj.backward()
        return obj.detach().reshape(-1)[0]           # a device scalar: the step reads the running loss back once, not per micro-batch

    def _precompute(self, local, n_leaves, batch_args):
        """Frozen trunk prefix: ONE batch construction for the whole local slice (same image order as micro-batch by micro-batch) and one
        prefix launch; the micro-batches then run suffix + head on their rows of the features."""
        if not (getattr(self.P, 'train_trunk_per_minibatch', True) and n_leaves >= 1 and getattr(self.net, 'trunk_precomputable', lambda: False)()):
            return None
        rng_state = random.getstate()
        inputs, targets = self.make_batch(local, len(local), **batch_args)
        feats = (self.net.precompute_trunk(*inputs, cache=True) if getattr(self.P, 'train_prefix_cache', False)
                 else self.net.precompute_trunk(*inputs))
        if feats is None:                            # (CPU tensors, ragged shapes): the batch is built again per micro-batch, from the same random state
            random.setstate(rng_state)
            return None
        return feats, targets

    def _leaves_batched(self, eng, head, mine, offsets, mini_size, pre, sink):
        """All local micro-batches through the trainable trunk suffix (isx/suffix.py) and -- when `head` is given -- the descriptor head
        (isx/head.py) in ONE forward and ONE backward each, their gradients kept apart per micro-batch.  Every row is computed exactly as in
        a launch of its own, per-leaf sums run over the leaf's rows / pixels in order: each leaf's flat gradient is bit for bit what a pass
        over that leaf alone produces -- and what a rank of a P-rank run computes.  The loss is the training script's own callback,
        evaluated per micro-batch on its rows.  Without `head` the head runs micro-batch by micro-batch under torch autograd on its rows of
        the suffix output.  Returns (flat_all (L, total), losses)."""
        P = self.P
        feats, targets_all = pre
        nb, k, L = len(feats), len(mine[0]), len(mine)
        rows = nb * k
        flat = self.flat.flat
        flat_all = torch.zeros((L, flat.numel()), dtype=flat.dtype, device=flat.device)
        with _phase("suffix_forward"):
            x_all = torch.cat([f[o:o + k] for o in offsets for f in feats], 0)      # leaf-major: [a_0; p_0; n_0; a_1; ...]
            if eng is not None:
                y_all, saved = eng.forward(x_all)                                    # no graph: the engine's backward is driven by hand below
            else:
                y_all, saved = x_all, None                                           # the whole trunk is frozen: nothing to train below the head
        losses = []

        def leaf_loss(out, j):
            o = offsets[j]
            targets = [t[o:o + k] if torch.is_tensor(t) and t.dim() > 0 and t.size(0) == feats[0].size(0) else t for t in targets_all]
            loss, loss2 = self.make_loss(out, targets)
            share = k / float(mini_size)
            obj = loss * share if P.train_loss_avg else loss
            if loss2 is not None:
                obj = obj + P.train_loss2_alpha * (loss2 * share if P.train_loss2_avg else loss2)
            obj.backward()
            losses.append(obj.detach().reshape(-1)[0])

        ph = _phase("heads")
        ph.__enter__()
        if head is not None:
            from model.siamese import _SplitRows
            shard = sink.shard_for(head.lin.weight) if sink is not None else None
            d_all, hctx = head.forward(y_all, shard=shard, leaf_ids=list(sink.leaf_ids) if shard is not None else None)
            trip = getattr(self.make_loss, 'triplet', None)
            if trip is not None and nb == 3 and d_all.is_cuda and getattr(P, 'train_loss_batched', True):
                # the training script declared its loss to be THE triplet criterion on (anchor, positive, negative) and nothing else
                # (`create_loss.triplet = criterion`): all L micro-batches in one launch (isx_triplet_leaves) -- per row the arithmetic of the
                # criterion's own kernels, the leaf's loss = its rows' losses in row order -- instead of L x (forward, sum, backward, four copies)
                from isx import ops
                share = k / float(mini_size)
                per_leaf, dd = ops.triplet_leaves(d_all, L, trip.margin, trip.normalized, (1.0 / k) if trip.size_average else 1.0,
                                                  share if P.train_loss_avg else 1.0)
                if trip.size_average:
                    per_leaf = per_leaf / k
                if P.train_loss_avg:
                    per_leaf = per_leaf * share
                losses.extend(per_leaf.unbind(0))
                dy_all = head.backward(hctx, dd, L, sink, flat_all, self.flat.slices)
                ph.__exit__()
                if eng is not None:
                    with _phase("suffix_backward"):
                        eng.backward(saved, dy_all, leaves=L, leaf_grads=(flat_all, self.flat.slices))
                return flat_all, losses
            dd = torch.empty_like(d_all)
            for j in range(L):
                d = d_all[j * rows:(j + 1) * rows].detach().requires_grad_(True)
                leaf_loss(_SplitRows.apply(d, nb), j)
                if d.grad is None:                       # a loss that does not touch this leaf's descriptors
                    dd[j * rows:(j + 1) * rows].zero_()
                else:
                    dd[j * rows:(j + 1) * rows].copy_(d.grad)
            dy_all = head.backward(hctx, dd, L, sink, flat_all, self.flat.slices)
        else:
            dy_all = torch.empty_like(y_all)
            for j in range(L):
                z = y_all[j * rows:(j + 1) * rows].detach().requires_grad_(True)
                leaf_loss(self.net.head_rows(z, nb), j)
                if z.grad is None:
                    dy_all[j * rows:(j + 1) * rows].zero_()
                else:
                    dy_all[j * rows:(j + 1) * rows].copy_(z.grad)
                self.flat.attach_all()
                flat_all[j].copy_(flat)                                              # the head's small parameters; the suffix slots are still zero
                flat.zero_()
        ph.__exit__()
        if eng is not None:
            with _phase("suffix_backward"):
                eng.backward(saved, dy_all, leaves=L, leaf_grads=(flat_all, self.flat.slices))
        return flat_all, losses

    def _slice(self, mini_batch):
        """This rank's micro-batches of a mini-batch: (leaves, first leaf index, one past the last, their items in order, row offset of each leaf)."""
        from isx import dp
        P, n = self.P, len(mini_batch)
        mb = P.train_micro_batch if 0 < P.train_micro_batch < n else n
        if self.mode == 'tree':
            leaves = [mini_batch[s:s + mb] for s in range(0, n, mb)]
            lo, hi = dp.rank_leaves(len(leaves), self.world, self.rank)
            mine = leaves[lo:hi]
        else:
            sl = mini_batch[(n * self.rank) // self.world:(n * (self.rank + 1)) // self.world]
            mine = [sl[s:s + mb] for s in range(0, len(sl), mb)]
            lo, hi = 0, len(mine)
        local = [t for leaf in mine for t in leaf]
        offsets = [sum(len(l) for l in mine[:j]) for j in range(len(mine))]
        return mine, lo, hi, local, offsets

    def begin_epoch(self, dataset, batch_args):
        """The epoch's item list, for the prefix look-ahead of _precompute_ahead (None: every step computes its own prefix)."""
        self._epoch_set, self._ahead = dataset, None

    def _precompute_ahead(self, start, n, n_leaves, batch_args):
        """Frozen trunk prefix for SEVERAL consecutive mini-batches in one launch (P.train_prefix_ahead of them, default 8).  The prefix of an
        image depends neither on the batch it rides in nor -- being frozen -- on the optimizer steps in between, so the features a step reads are
        bit for bit the ones its own launch would produce; 192 images fill the chip to 0.67 of the fp32 peak on these layers, 768 to 0.8.  Only
        when the training script declares its batch construction deterministic (`create_batch.deterministic`: no random augmentation, negatives
        drawn per epoch) -- the mini-batches further down the list are built ahead of their turn."""
        P = self.P
        # P.train_prefix_ahead counts mini-batches of ONE process; a rank of a P-rank run holds 1/P of each, so it looks P times further ahead:
        # the launch carries the same number of images whatever the world size (each rank decides for itself: no collective is involved)
        G = int(getattr(P, 'train_prefix_ahead', 8))
        G = G * self.world if G > 1 else G
        ds = getattr(self, '_epoch_set', None)
        if (G < 2 or ds is None or start is None or n_leaves < 1 or not getattr(self.make_batch, 'deterministic', False)
                or getattr(P, 'train_prefix_cache', False) or not getattr(P, 'train_trunk_per_minibatch', True)
                or not getattr(self.net, 'trunk_precomputable', lambda: False)()):
            return None
        a = self._ahead
        if a is None or not (a[0] <= start < a[0] + a[1] * n) or (start - a[0]) % n:
            groups, s_ = [], start
            while len(groups) < G and s_ + n <= len(ds):
                groups.append(self._slice(ds[s_:s_ + n])[3])
                s_ += n
            if len(groups) < 2 or len(set(len(g) for g in groups)) != 1 or not groups[0]:
                self._ahead = None
                return None
            items = [t for g in groups for t in g]
            inputs, targets = self.make_batch(items, len(items), **batch_args)
            feats = self.net.precompute_trunk(*inputs)
            if feats is None:
                self._epoch_set = None           # CPU tensors / ragged shapes: the steps build their own batches
                return None
            a = self._ahead = (start, len(groups), len(groups[0]), feats, targets, len(items))
        g, m = (start - a[0]) // n, a[2]
        feats = tuple(f[g * m:(g + 1) * m] for f in a[3])
        targets = [t[g * m:(g + 1) * m] if torch.is_tensor(t) and t.dim() > 0 and t.size(0) == a[5] else t for t in a[4]]
        if g == a[1] - 1:
            self._ahead = None                   # last user: the block is freed with this step
        return feats, targets

    def step(self, optimizer, mini_batch, batch_args, start=None):
        from isx import dp
        P, n = self.P, len(mini_batch)
        mb = P.train_micro_batch if 0 < P.train_micro_batch < n else n
        mine, lo, hi, local, offsets = self._slice(mini_batch)
        with _phase("batch+prefix"):
            pre = self._precompute_ahead(start, n, len(mine), batch_args) if mine else None
            if pre is None:
                pre = self._precompute(local, len(mine), batch_args) if mine else None
        self.flat.zero_grad()
        losses = []
        fuse = optimizer if getattr(P, 'train_fused_head_sgd', True) else None       # head weight: gradient + SGD update as one kernel (isx/dp.py)
        # sharded head this step?  decided from what EVERY rank knows (mini-batch size, micro-batch size, world): equal micro-batches, the same
        # number on every rank (GPU) / a multiple of the 8 feature groups' rank counts (CPU: one rule for every world size, see __init__)
        n_leaves = -(-n // mb)
        shard_ok = bool(self.shards) and n % mb == 0 and self.mode == 'tree' and n_leaves % (self.world if next(iter(self.shards.values())).weight.is_cuda else 8) == 0
        if self.shards and not shard_ok:
            self.sync_head()                          # the replicated path below reads and updates the whole weight
        with dp.RowSink(self.deferred, optimizer=(optimizer if shard_ok else fuse), shards=(self.shards if shard_ok else None)) as sink:
            for sh in sink.shards.values():
                sh.begin_step(len(mine))
            sink.leaf_ids = list(range(lo, hi))
            eng = head = None
            batched = getattr(self.P, 'train_suffix_batched', True)
            # batched engines need equal micro-batches; decided from the MINI-BATCH (n % mb == 0: every leaf of every rank has mb triplets), not from
            # this rank's leaves, so that every rank and the single-process run take the same numeric path (round-4 ADVICE)
            if self.mode == 'tree' and pre is not None and mine and n % mb == 0 and batched:
                eng = getattr(self.net, 'suffix_engine', lambda: None)()
                head = getattr(self.net, 'head_engine', lambda: None)()
            if eng is not None or head is not None:
                if batched == "leaf":                # test mode: the same machinery one micro-batch at a time (what a rank with ONE leaf runs)
                    parts = [self._leaves_batched(eng, head, [mine[j]], [offsets[j]], n, pre, sink) for j in range(len(mine))]
                    flat_all = torch.cat([p[0] for p in parts], 0)
                    losses = [l for p in parts for l in p[1]]
                else:
                    flat_all, losses = self._leaves_batched(eng, head, mine, offsets, n, pre, sink)
                if flat_all.is_cuda and 1 <= hi - lo <= 16:
                    from isx import ops
                    self.flat.attach_all()
                    ops.tree_sum_rows(flat_all, out=self.flat.flat)       # the same tree, one pass over the leaves' rows instead of L - 1 add passes
                else:
                    self.flat.put(dp.tree_sum(lo, hi, lambda i: flat_all[i - lo]))
                if self.exchange is not None:
                    self.exchange.allreduce_(self.flat.flat)
            elif self.mode == 'tree':
                def leaf(i):
                    sink.leaf_ids = [i]
                    losses.append(self._forward_backward(mine[i - lo], offsets[i - lo], n, batch_args, pre))
                    return self.flat.take()
                self.flat.put(dp.tree_sum(lo, hi, leaf))
                if self.exchange is not None:
                    self.exchange.allreduce_(self.flat.flat)
            else:
                for j, triplets in enumerate(mine):
                    losses.append(self._forward_backward(triplets, offsets[j], n, batch_args, pre, arm=(j == len(mine) - 1)))
                self.flat.finish()
            with _phase("head_weight_gradient"):
                sink.finish()
        loss = torch.stack(losses).double().sum() if losses else torch.zeros((), dtype=torch.float64)
        if self.world > 1:
            t = loss.reshape(1).to(next(self.net.parameters()).device)
            dist.all_reduce(t)
            loss = t
        with _phase("optimizer"):
            optimizer.step()
        # the loss stays a tensor (float64 scalar, on the device of the step): output_stats adds it to the running loss there and reads that back
        # when a line is printed (every P.train_loss_int steps) -- a read-back per step would leave the GPU idle while the host prepares the next
        loss = loss.reshape(())
        if self.world > 1:
            # BatchNorm in training mode (P.train_bn): each rank's running statistics saw only its slice -- average them so that the
            # replicas stay one model (same mining, same evaluation, a checkpoint that is every rank's)
            from isx.dp import average_buffers, batch_norm_buffers
            average_buffers(batch_norm_buffers(self.net))
        return loss


def train_gen(train_type, P, test_print, test_net, net, train_set, testset_tuple, optimizer, create_epoch, create_batch, create_loss,
              best_score=0):
    """Generic training driver (reference utils/train_general.py:77-105): per epoch anneal -> create_epoch ->
    mini-batches of P.train_batch_size (a trailing partial one is dropped) -> statistics / evaluation."""
    set_net_train(net, True, bn_train=P.train_bn)
    rank, world = _dp()
    if world > 1:
        from isx.dp import broadcast_module_state
        broadcast_module_state(net, src=0)      # replicas start from rank 0's weights and buffers (the descriptor head is random-init)
    stepper = _Stepper(P, net, create_batch, create_loss)
    for epoch in range(P.train_epochs):
        optimizer = anneal(net, optimizer, epoch, P.train_annealing)
        if world > 1 or getattr(P, 'train_seed', None) is not None:
            # identical couple order (and random fall-back negatives) on every rank -- and, with P.train_seed set, in a single process,
            # so that runs with 1, 2, 4, 8 ranks train on the same triplets (the reference leaves `random` unseeded)
            random.seed((getattr(P, 'train_seed', 0) or 0) + epoch)
        stepper.sync_head()                     # the epoch's embedding pass reads the whole head
        dataset, batch_args = create_epoch(epoch, train_set, testset_tuple)
        stepper.begin_epoch(dataset, batch_args)

        def one(state, start, is_final, mini_batch):
            count, score, running = state
            loss = stepper.step(optimizer, mini_batch, batch_args, start=start)
            t = P.train_test_int
            if (t > 0 and count % t == t - 1) or (t <= 0 and is_final):
                stepper.sync_head()             # output_stats evaluates the net now
            running, score = output_stats(train_type, P, test_print, test_net, net, testset_tuple, epoch, count, is_final, loss,
                                          running, score)
            return count + 1, score, running

        _, best_score, _ = fold_batches(one, (0, best_score, 0.0), dataset, P.train_batch_size, cut_end=True)
    stepper.sync_head()
    return best_score
