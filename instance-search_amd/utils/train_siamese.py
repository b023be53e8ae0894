"""Descriptor-net evaluation helpers with the reference's names
(utils/train_siamese.py: embeddings_device_dim :30-43, get_similarities :48-55,
test_descriptor_net :61-82).  The similarity matrix stays on the GPU: cosine via libisx
`isx_cosine_sim`, the label-masked sums via `isx_masked_sums` instead of the reference's
O(M*N) Python generator."""
import os
import random

import torch
import torch.distributed as dist

from model.nn_utils import set_net_train
from .dataset import get_pos_couples
from .general import is_main_process, log
from .metrics import _label_ids, mean_avg_precision, precision1


def embeddings_device_dim(P, net, n, sim_matrix=False):
    """(device, descriptor width).  The reference moves the slab -- or, with sim_matrix, everything -- to the CPU once it
    exceeds P.embeddings_cuda_size bytes (2**30 there, utils/train_siamese.py:30-43).  Here only a SLAB beyond the budget
    (sized for 288 GB of HBM) leaves the GPU; an n x n score matrix beyond the budget is never built: its consumers work on
    query-row blocks (utils.metrics.retrieval_metrics, SimilarityRows below), same values, bounded memory."""
    device, out_size = P.cuda_device, P.feature_dim
    if hasattr(net, 'feature_size') and out_size <= 0:
        out_size = net.feature_size
    if n * out_size * 4 > P.embeddings_cuda_size:
        device = -1
    return device, out_size


class SimilarityRows(object):
    """A descriptor slab standing in for its own n x n similarity matrix when that matrix is over budget: `rows(r0, r1)`
    computes one block of rows on demand (isx_cosine_sim on the GPU)."""

    def __init__(self, emb):
        self.emb = emb
        self.is_cuda = emb.is_cuda

    def size(self, dim=None):
        n = self.emb.size(0)
        return (n, n) if dim is None else n

    def rows(self, r0, r1):
        return similarity_matrix(self.emb[r0:r1], self.emb)


def similarity_matrix(a, b):
    """a @ b.T for unit-norm descriptor slabs (reference: torch.mm, test/*_test.py)."""
    if a.is_cuda:
        from isx import ops
        return ops.cosine_sim(a.float(), b.float())
    return torch.mm(a, b.t())


def dp_get_embeddings(get_embeddings, net, dataset, device, out_size):
    """get_embeddings under data-parallel TRAINING: on the GPU every rank extracts a contiguous 1/P of the set and the descriptor rows are
    all-gathered (the per-epoch embedding pass and the evaluations in between were run whole by every rank -- P times the work for the same
    tensor).  The HIP path gives an image the same descriptor whatever batch it rides in, so the gathered slab is the single-process one bit for
    bit: same mined negatives, same metrics on every rank.  CPU runs (torch's convolutions may pick kernels by batch size) keep the whole pass."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world == 1 or device < 0 or len(dataset) < world:
        return get_embeddings(net, dataset, device, out_size)
    rank = dist.get_rank()
    bounds = [((len(dataset) * r) // world, (len(dataset) * (r + 1)) // world) for r in range(world)]
    lo, hi = bounds[rank]
    local = get_embeddings(net, dataset[lo:hi], device, out_size)
    most = max(b - a for a, b in bounds)
    pad = local.new_zeros((most, local.size(1)))
    pad[:local.size(0)].copy_(local)
    if dist.get_backend() == 'nccl':
        out = pad.new_empty((world * most, local.size(1)))
        dist.all_gather_into_tensor(out, pad)
        parts = [out[r * most:r * most + (b - a)] for r, (a, b) in enumerate(bounds)]
    else:
        host = pad.cpu()
        bufs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(bufs, host)
        parts = [bufs[r][:b - a].to(local.device) for r, (a, b) in enumerate(bounds)]
    return torch.cat(parts, 0)


def get_similarities(P, get_embeddings, net, dataset):
    """(n x n similarities of the dataset's descriptors, device).  Over utils.metrics.SIM_BUDGET_BYTES the matrix is
    returned as a SimilarityRows (row blocks on demand) -- train.siamese_descriptor.mine_epoch_negatives consumes both."""
    from . import metrics
    set_net_train(net, False)
    d, o = embeddings_device_dim(P, net, len(dataset), sim_matrix=True)
    emb = dp_get_embeddings(get_embeddings, net, dataset, d, o)
    n = emb.size(0)
    sim = SimilarityRows(emb) if n * n * 4 > metrics.SIM_BUDGET_BYTES else similarity_matrix(emb, emb)
    set_net_train(net, True, bn_train=P.train_bn)
    return sim, d


def test_descriptor_net(P, get_embeddings, net, test_set, test_ref_set, kth=1):
    from .metrics import retrieval_metrics
    d, o = embeddings_device_dim(P, net, max(len(test_set), len(test_ref_set)))
    m = retrieval_metrics(dp_get_embeddings(get_embeddings, net, test_set, d, o), dp_get_embeddings(get_embeddings, net, test_ref_set, d, o),
                          test_set, test_ref_set, kth, with_sums=True)
    sum_neg = m['sum_all'] - m['sum_pos']
    sum_max = float(m['max_sim'].double().sum())
    lab_dict = dict((lab, {}) for _, lab, _ in test_set)
    for (_, lab, _), got in zip(test_set, m['max_label']):
        seen = lab_dict[lab]
        seen.setdefault(got, seen.get(got, 0) + 1)
    return m['prec1'], m['correct'], m['total'], m['sum_pos'], sum_neg, sum_max, m['mAP'], lab_dict


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
    # data parallel: every rank evaluates (same weights, same result, same control flow), rank 0 alone writes
    main = is_main_process()
    if correct > best_score:
        best_score = correct
        if save_dir and main:
            torch.save(net.state_dict(), os.path.join(save_dir, 'best_siam.pth.tar'))
    if save_dir and main:
        torch.save(net.state_dict(), os.path.join(save_dir, 'model_siam_' + str(epoch) + '.pth.tar'))
    if save_dir and dist.is_available() and dist.is_initialized():
        dist.barrier()                      # nobody races ahead of (or reads) a half-written checkpoint
    couples = get_pos_couples(test_ref_set)
    sample = random.sample(test_ref_set, max(1, len(test_ref_set) // 10))
    sample = [x for x in sample if len(couples.get(x[1], ())) >= 3]
    if sample:
        evaluate('TRAIN - ', sample, test_ref_set, 2)
    set_net_train(net, True, bn_train=P.train_bn)
    return best_score
