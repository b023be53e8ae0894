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


def micro_batch_gen(last, i, is_final, batch, P, net, create_batch, batch_args, create_loss, reducer=None):
    prev_val, mini_batch_size = last
    n = len(batch)
    tensors_in, labels_in = create_batch(batch, n, **batch_args)
    tensors_out = net(*tensors_in)
    loss, loss2 = create_loss(tensors_out, labels_in)
    loss_micro = loss * n / mini_batch_size if P.train_loss_avg else loss
    val = float(loss_micro.detach().reshape(-1)[0])
    if loss2 is not None:
        loss2_micro = loss2 * n / mini_batch_size if P.train_loss2_avg else loss2
        loss_micro = loss_micro + P.train_loss2_alpha * loss2_micro
        val += P.train_loss2_alpha * float(loss2_micro.detach().reshape(-1)[0])
    if reducer is not None and is_final:
        reducer.arm()                                # overlap the bucketed all-reduce with this last backward
    loss_micro.backward()
    return prev_val + val, mini_batch_size


def mini_batch_gen(last, i, is_final, batch, train_type, P, test_print, test_net, net, optimizer, testset_tuple, epoch, micro_args,
                   reducer=None):
    batch_count, score, running_loss = last
    rank, world = _dp()
    if reducer is not None:
        reducer.zero_grad()
    else:
        optimizer.zero_grad()
    lo, hi = (len(batch) * rank) // world, (len(batch) * (rank + 1)) // world
    mine = batch[lo:hi]
    args = dict(micro_args)
    args['reducer'] = reducer
    loss, _ = fold_batches(micro_batch_gen, (0.0, len(batch)), mine, P.train_micro_batch, add_args=args) if mine else (0.0, len(batch))
    if reducer is not None:
        reducer.finish()
    if world > 1:
        t = torch.tensor([loss], dtype=torch.float64, device=next(net.parameters()).device)
        dist.all_reduce(t)
        loss = float(t.item())
    optimizer.step()
    running_loss, score = output_stats(train_type, P, test_print, test_net, net, testset_tuple, epoch, batch_count, is_final, loss,
                                       running_loss, score)
    return batch_count + 1, score, running_loss


def train_gen(train_type, P, test_print, test_net, net, train_set, testset_tuple, optimizer, create_epoch, create_batch, create_loss,
              best_score=0):
    set_net_train(net, True, bn_train=P.train_bn)
    rank, world = _dp()
    reducer = None
    if world > 1:
        from isx.dp import GradAllReducer
        reducer = GradAllReducer(list(net.parameters()))
    for epoch in range(P.train_epochs):
        optimizer = anneal(net, optimizer, epoch, P.train_annealing)
        if world > 1:
            random.seed(getattr(P, 'train_seed', 0) + epoch)    # identical couple order on every rank
        dataset, batch_args = create_epoch(epoch, train_set, testset_tuple)
        micro_args = {'P': P, 'net': net, 'create_batch': create_batch, 'batch_args': batch_args, 'create_loss': create_loss}
        mini_args = {'train_type': train_type, 'P': P, 'test_print': test_print, 'test_net': test_net, 'net': net,
                     'optimizer': optimizer, 'testset_tuple': testset_tuple, 'epoch': epoch, 'micro_args': micro_args,
                     'reducer': reducer}
        _, best_score, _ = fold_batches(mini_batch_gen, (0, best_score, 0.0), dataset, P.train_batch_size, cut_end=True, add_args=mini_args)
    return best_score
