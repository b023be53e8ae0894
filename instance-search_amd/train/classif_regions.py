"""Sub-region classifier approach: get_embeddings (reference train/classif_regions.py:107-132),
get_class_net (:135-149), test_classif_net (:25-51).  Descriptor = the class-score vector at the
location whose best class score is highest, L2-normalised."""
import torch

from model.custom_modules import l2_normalize_rows
from model.siamese import TuneClassif, TuneClassifSub
from utils import move_device, tensor
from ._common import base_model, device_batch_size, fold_shape_buckets, label_index, load_weights, make_resident, scatter_rows, test_transform
from .classif_regions_p import P

labels = []


def _best_location_descriptors(score_map):
    """(B, n_cls, H', W') -> (B, n_cls): libisx `isx_best_location_desc` on the GPU."""
    if score_map.is_cuda:
        from isx import ops
        return ops.best_location_desc(score_map.float())[0]      # NCHW or channels-last, consumed in place
    B, K, Hp, Wp = score_map.shape
    mx = score_map.max(1)[0]                                   # class-max map
    # first maximal index: smallest column, then smallest row in that column
    colmax, row_of_col = mx.max(1)                             # over rows, per column
    col = colmax.max(1)[1]
    row = row_of_col.gather(1, col[:, None])[:, 0]
    picked = score_map[torch.arange(B), :, row, col]
    return l2_normalize_rows(picked)


def test_classif_net(net, test_set):
    """(correct, total): the prediction of an image is the class of its globally highest score over all locations.  The reference walks the set
    one image at a time (:50, "batch size has to be 1 here": sizes may differ); an image's score map does not depend on the batch it rides in,
    so same-shaped images share a launch here (fold_shape_buckets) -- 15 ms of host work per single-image pass otherwise."""
    trans = test_transform(P)
    if trans is None:
        make_resident(test_set, P.cuda_device)          # the queries are classified here AND embedded right after: uploaded once
    ids = label_index(labels)
    correct = [0]

    def run(indices, batch, x):
        with torch.no_grad():
            out = net(x)[0]
            pred = out.max(1)[0].flatten(1).argmax(1)
            flat = out.flatten(2)
            cls = flat[torch.arange(out.size(0), device=out.device), :, pred].argmax(1).tolist()
        correct[0] += sum(1 for (_, lab, _), p in zip(batch, cls) if ids[lab] == p)

    fold_shape_buckets(run, test_set, lambda shape: device_batch_size(P, test_set, shape), stage=(trans, P.cuda_device))
    return correct[0], len(test_set)


def get_embeddings(net, dataset, device, out_size):
    trans = test_transform(P)
    if trans is None:
        make_resident(dataset, P.cuda_device)
    slab = tensor(device, len(dataset), out_size)

    def run(indices, batch, x):
        with torch.no_grad():
            out = net(x)[0]
            scatter_rows(slab, indices, _best_location_descriptors(out))

    # the reference walks one image at a time (images may differ in size); here images are bucketed by shape and every
    # bucket goes through in batches of P.test_batch_size, staged ahead of the trunk (BatchStager)
    fold_shape_buckets(run, dataset, lambda shape: device_batch_size(P, dataset, shape), stage=(trans, P.cuda_device))
    return slab


def get_class_net():
    if P.bn_model:
        bn_model = load_weights(TuneClassif(base_model(P, pretrained=False), len(labels)), P.bn_model)
    else:
        bn_model = base_model(P)
    net = TuneClassifSub(bn_model, len(labels), P.feature_size2d, untrained=P.untrained_blocks)
    return move_device(load_weights(net, P.preload_net), P.cuda_device)
