"""Descriptor-net evaluation helpers with the reference's names
(utils/train_siamese.py: embeddings_device_dim :30-43, get_similarities :48-55,
test_descriptor_net :61-82).  The similarity matrix stays on the GPU: cosine via libisx
`isx_cosine_sim`, the label-masked sums via `isx_masked_sums` instead of the reference's
O(M*N) Python generator."""
import os
import random

import torch

from model.nn_utils import set_net_train
from .dataset import get_pos_couples
from .general import log
from .metrics import _label_ids, mean_avg_precision, precision1


def embeddings_device_dim(P, net, n, sim_matrix=False):
    """(device, descriptor width): the configured GPU unless the slab (or the n x n matrix)
    exceeds P.embeddings_cuda_size bytes."""
    device, out_size = P.cuda_device, P.feature_dim
    if hasattr(net, 'feature_size') and out_size <= 0:
        out_size = net.feature_size
    if n * out_size * 4 > P.embeddings_cuda_size:
        device = -1
    if sim_matrix and n * n * 4 > P.embeddings_cuda_size:
        device = -1
    return device, out_size


def similarity_matrix(a, b):
    """a @ b.T for unit-norm descriptor slabs (reference: torch.mm, test/*_test.py)."""
    if a.is_cuda:
        from isx import ops
        return ops.cosine_sim(a.float(), b.float())
    return torch.mm(a, b.t())


def get_similarities(P, get_embeddings, net, dataset):
    set_net_train(net, False)
    d, o = embeddings_device_dim(P, net, len(dataset), sim_matrix=True)
    emb = get_embeddings(net, dataset, d, o)
    sim = similarity_matrix(emb, emb)
    set_net_train(net, True, bn_train=P.train_bn)
    return sim, d


def test_descriptor_net(P, get_embeddings, net, test_set, test_ref_set, kth=1):
    d, o = embeddings_device_dim(P, net, max(len(test_set), len(test_ref_set)))
    sim = similarity_matrix(get_embeddings(net, test_set, d, o), get_embeddings(net, test_ref_set, d, o))
    prec1, correct, total, max_sim, max_label = precision1(sim, test_set, test_ref_set, kth)
    mAP = mean_avg_precision(sim, test_set, test_ref_set, kth)
    qlab, glab = _label_ids(test_set, test_ref_set)
    if sim.is_cuda:
        from isx import ops
        rows = ops.masked_sums(sim, qlab.cuda(), glab.cuda()).cpu()
        sum_pos, sum_all = float(sum(rows[:, 0].tolist())), float(sum(rows[:, 1].tolist()))
    else:
        mask = qlab[:, None] == glab[None, :]
        sum_pos, sum_all = float(sim[mask].double().sum()), float(sim.double().sum())
    sum_neg = sum_all - sum_pos
    sum_max = float(max_sim.double().sum())
    lab_dict = dict((lab, {}) for _, lab, _ in test_set)
    for (_, lab, _), got in zip(test_set, max_label):
        seen = lab_dict[lab]
        seen.setdefault(got, seen.get(got, 0) + 1)
    return prec1, correct, total, sum_pos, sum_neg, sum_max, mAP, lab_dict


def test_print_descriptor(train_type, P, net, testset_tuple, get_embeddings, best_score=0, epoch=0):
    """Evaluate as a descriptor net on (test_set, test_ref_set), then on a 10 % sample of the reference
    set against itself with kth = 2 (reference utils/train_siamese.py:86-120); logs both lines, saves
    the weights when P.save_dir is set, returns the best score."""
    def stats(prefix, p1, c, t, avg_pos, avg_neg, avg_max, mAP):
        log(P, prefix + 'Correct: {0} / {1} - acc: {2:.4f} - mAP:{3:.4f}\n'.format(c, t, p1, mAP) +
            'AVG cosine sim (sq dist) values: pos: {0:.4f} ({1:.4f}), neg: {2:.4f} ({3:.4f}), max: {4:.4f} ({5:.4f})'.format(
                avg_pos, 2 - 2 * avg_pos, avg_neg, 2 - 2 * avg_neg, avg_max, 2 - 2 * avg_max))

    def evaluate(prefix, queries, gallery, kth):
        p1, c, t, s_pos, s_neg, s_max, mAP, _ = test_descriptor_net(P, get_embeddings, net, queries, gallery, kth)
        n_pos = sum(1 for _, a, _ in queries for _, b, _ in gallery if a == b)
        n_neg = len(queries) * len(gallery) - n_pos
        stats(prefix, p1, c, t, s_pos / max(n_pos, 1), s_neg / max(n_neg, 1), s_max / max(len(queries), 1), mAP)
        return c

    test_set, test_ref_set = testset_tuple
    set_net_train(net, False)
    correct = evaluate('TEST - ', test_set, test_ref_set, 1)
    save_dir = getattr(P, 'save_dir', None)
    if correct > best_score:
        best_score = correct
        if save_dir:
            torch.save(net.state_dict(), os.path.join(save_dir, 'best_siam.pth.tar'))
    if save_dir:
        torch.save(net.state_dict(), os.path.join(save_dir, 'model_siam_' + str(epoch) + '.pth.tar'))
    couples = get_pos_couples(test_ref_set)
    sample = random.sample(test_ref_set, max(1, len(test_ref_set) // 10))
    sample = [x for x in sample if len(couples.get(x[1], ())) >= 3]
    if sample:
        evaluate('TRAIN - ', sample, test_ref_set, 2)
    set_net_train(net, True, bn_train=P.train_bn)
    return best_score
