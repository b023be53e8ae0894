"""Descriptor-net evaluation helpers with the reference's names
(utils/train_siamese.py: embeddings_device_dim :30-43, get_similarities :48-55,
test_descriptor_net :61-82).  The similarity matrix stays on the GPU: cosine via libisx
`isx_cosine_sim`, the label-masked sums via `isx_masked_sums` instead of the reference's
O(M*N) Python generator."""
import torch

from model.nn_utils import set_net_train
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
