"""Batch fold driver and the generic training loop (reference utils/train_general.py): fold_batches
(:27-38), anneal (:41-48), micro/mini-batch steps with gradient accumulation (:51-74), train_gen
(:77-105), output_stats (:12-23).  Added: data-parallel execution -- with torch.distributed
initialised every rank takes an equal slice of each mini-batch and the summed gradients are
all-reduced over RCCL before the optimizer step (isx.dp.GradAllReducer)."""
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
    running_loss += loss
    if batch_count % every == every - 1:
        log(P, '[{0:d}, {1:5d}] loss: {2:.5f}'.format(epoch + 1, batch_count + 1, running_loss / every))
        running_loss = 0.0
    t = P.train_test_int
    if (t > 0 and batch_count % t == t - 1) or (t <= 0 and is_final):
        score = test_print(train_type, P, net, testset_tuple, test_net, score, epoch + 1)
    return running_loss, score


def anneal(net, optimizer, epoch, annealing_dict):
    """At the epochs listed in annealing_dict a NEW SGD is built with lr scaled by the given factor
    (momentum buffers start afresh, as in the reference)."""
    if epoch not in annealing_dict:
        return optimizer
    g = optimizer.state_dict()['param_groups'][0]
    return optim.SGD((p for p in net.parameters() if p.requires_grad), lr=g['lr'] * annealing_dict[epoch],
                     momentum=g['momentum'], weight_decay=g['weight_decay'])


def _dp():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


class _Stepper(object):
    """One optimizer step = gradient accumulation over the micro-batches of this rank's slice of a mini-batch
    (reference utils/train_general.py:51-74: micro_batch_gen / mini_batch_gen), then -- data parallel -- the
    bucketed gradient all-reduce, overlapped with the backward of the last micro-batch."""

    def __init__(self, P, net, make_batch, make_loss, reducer):
        self.P, self.net, self.make_batch, self.make_loss, self.reducer = P, net, make_batch, make_loss, reducer
        self.rank, self.world = _dp()

    def _accumulate(self, total, start, is_last, triplets, mini_size, batch_args, pre=None):
        P = self.P
        if pre is not None:
            # trunk features of the whole slice were computed in one launch (frozen trunk): this micro-batch takes its rows
            feats, targets_all = pre
            n = len(triplets)
            out = self.net.forward_features(*[f[start:start + n] for f in feats])
            targets = [t[start:start + n] if torch.is_tensor(t) and t.dim() > 0 and t.size(0) == feats[0].size(0) else t for t in targets_all]
            loss, loss2 = self.make_loss(out, targets)
        else:
            inputs, targets = self.make_batch(triplets, len(triplets), **batch_args)
            loss, loss2 = self.make_loss(self.net(*inputs), targets)
        return self._backward(total, is_last, loss, loss2, len(triplets), mini_size)

    def _backward(self, total, is_last, loss, loss2, k, mini_size):
        P = self.P
        share = k / float(mini_size)
        obj = loss * share if P.train_loss_avg else loss
        if loss2 is not None:
            obj = obj + P.train_loss2_alpha * (loss2 * share if P.train_loss2_avg else loss2)
        if self.reducer is not None and is_last:
            self.reducer.arm()                       # exchange buckets as this last backward fills them
        obj.backward()
        return total + obj.detach().reshape(-1)[0]    # a device scalar: the step reads the running loss back once, not per micro-batch

    def step(self, optimizer, mini_batch, batch_args):
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            optimizer.zero_grad()
        n = len(mini_batch)
        mine = mini_batch[(n * self.rank) // self.world:(n * (self.rank + 1)) // self.world]
        loss = 0.0
        if mine:
            pre = None
            if (getattr(self.P, 'train_trunk_per_minibatch', True) and 0 < self.P.train_micro_batch < len(mine)
                    and getattr(self.net, 'trunk_precomputable', lambda: False)()):
                # frozen trunk: ONE batch construction for the whole slice (same image / random-negative order as micro-batch by micro-batch)
                # and one trunk launch; the micro-batches below run the head on their rows of the features
                rng_state = random.getstate()
                inputs, targets = self.make_batch(mine, len(mine), **batch_args)
                feats = self.net.precompute_trunk(*inputs)
                if feats is not None:
                    pre = (feats, targets)
                else:                                # (CPU tensors, ragged shapes): the batch is built again per micro-batch, from the same random state
                    del inputs, targets
                    random.setstate(rng_state)
            loss = fold_batches(self._accumulate, 0.0, mine, self.P.train_micro_batch,
                                add_args={'mini_size': n, 'batch_args': batch_args, 'pre': pre})
        if self.reducer is not None:
            self.reducer.finish()
        if self.world > 1:
            t = (loss.double() if torch.is_tensor(loss) else torch.tensor(loss, dtype=torch.float64)).reshape(1).to(next(self.net.parameters()).device)
            dist.all_reduce(t)
            loss = t
        optimizer.step()
        loss = float(loss)                            # the one read-back of the step (after the optimizer has been enqueued)
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
    reducer = None
    if world > 1:
        from isx.dp import GradAllReducer, broadcast_module_state
        broadcast_module_state(net, src=0)      # replicas start from rank 0's weights and buffers (the descriptor head is random-init)
        reducer = GradAllReducer(list(net.parameters()))
    stepper = _Stepper(P, net, create_batch, create_loss, reducer)
    for epoch in range(P.train_epochs):
        optimizer = anneal(net, optimizer, epoch, P.train_annealing)
        if world > 1:
            random.seed(getattr(P, 'train_seed', 0) + epoch)    # identical couple order on every rank
        dataset, batch_args = create_epoch(epoch, train_set, testset_tuple)

        def one(state, start, is_final, mini_batch):
            count, score, running = state
            loss = stepper.step(optimizer, mini_batch, batch_args)
            running, score = output_stats(train_type, P, test_print, test_net, net, testset_tuple, epoch, count, is_final, loss,
                                          running, score)
            return count + 1, score, running

        _, best_score, _ = fold_batches(one, (0, best_score, 0.0), dataset, P.train_batch_size, cut_end=True)
    return best_score
