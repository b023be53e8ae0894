"""Global-descriptor approach: the retrieval-path functions of the reference's
train/classif_finetune.py -- get_embeddings (:82-110), get_class_net (:113-121),
test_classif_net (:25-50).  The fine-tuning loop (:53-78, main) is out of scope."""
import torch
import torch.nn as nn

from model.custom_modules import l2_normalize_rows
from model.siamese import TuneClassif
from utils import fold_batches, move_device, tensor
from ._common import BatchStager, base_model, device_batch_size, label_index, load_weights, make_resident, stage_batch, test_transform
from .classif_finetune_p import P

labels = []   # filled by the entry point once the reference set is listed, then constant


def test_classif_net(net, test_set):
    """(correct, total) classification accuracy of an eval-mode net."""
    trans = test_transform(P)
    if trans is None:
        make_resident(test_set, P.cuda_device)          # get_embeddings needs the set in HBM right after: upload it once, here

    def run(acc, i, is_final, batch):
        correct, total = acc
        with torch.no_grad():
            pred = net(stage_batch(batch, trans, P.cuda_device)).argmax(1).tolist()
        ids = label_index(labels)
        correct += sum(1 for (_, lab, _), p in zip(batch, pred) if ids[lab] == p)
        return correct, total + len(batch)

    return fold_batches(run, (0, 0), test_set, device_batch_size(P, test_set))


def _full_map_pool(net, fmap):
    """True when feature_reduc is exactly one average pool spanning the whole feature map."""
    reduc = list(net.feature_reduc)
    if len(reduc) != 1 or not isinstance(reduc[0], nn.AvgPool2d):
        return False
    ks = reduc[0].kernel_size
    ks = ks if isinstance(ks, tuple) else (ks, ks)
    return tuple(fmap.shape[2:]) == tuple(ks)


def fc7_tap(classifier):
    """classifier[:6] of an AlexNet-style head -- (Dropout, Linear, ReLU, Dropout, Linear, ReLU | Linear(nbClass)), reference
    model/ModelDefinition.py:31-37 -- i.e. the 4096-d "fc7" activation; raises for heads without that layout (ResNet: one Linear)."""
    mods = list(classifier)
    if len(mods) < 7 or not (isinstance(mods[1], nn.Linear) and isinstance(mods[4], nn.Linear) and isinstance(mods[5], nn.ReLU)):
        raise ValueError('fc7 descriptors need an AlexNet-style classifier (Dropout, Linear, ReLU, Dropout, Linear, ReLU, Linear)')
    return nn.Sequential(*mods[:6])


def fc7_size(classifier):
    return fc7_tap(classifier)[4].out_features


def get_embeddings(net, dataset, device, out_size):
    """(len(dataset), out_size) slab of L2-normalised descriptors on `device`.
    P.embeddings_classify False: pooled convolutional features (classifier stripped for the
    pass, restored afterwards); True: the class scores; P.embeddings_fc7 (extension, AlexNet): classifier[:6].  On the GPU the pool + L2 of a batch is
    one fused kernel (`isx_gap_l2`) writing straight into the slab rows."""
    trans = test_transform(P)
    if trans is None:
        make_resident(dataset, P.cuda_device)           # the set goes to HBM once; batches are device-side row gathers
    fc7 = bool(getattr(P, 'embeddings_fc7', False)) and not P.embeddings_classify
    stripped = not P.embeddings_classify and not fc7
    classifier = net.classifier
    if stripped:
        net.classifier = nn.Sequential()
    elif fc7:
        net.classifier = fc7_tap(classifier)                # extension: the 4096-d activation behind the second ReLU (eval mode: Dropout = identity)
    slab = tensor(device, len(dataset), out_size)
    bs = device_batch_size(P, dataset)
    stager = BatchStager(dataset, bs, trans, P.cuda_device)    # resident: row gathers; beyond the HBM budget: double-buffered pinned staging on a copy stream

    def run(slab, i, is_final, batch):
        x = stager.get(i, batch)
        rows = slab[i:i + len(batch)]
        with torch.no_grad():
            if stripped and x.is_cuda and slab.is_cuda:
                fmap = net.features(x)
                if _full_map_pool(net, fmap):
                    from isx import ops
                    ops.gap_l2(fmap.float(), out=rows)        # NCHW or channels-last, consumed in place
                    return slab
                out = net.feature_reduc(fmap)
                out = out.view(out.size(0), -1)
            else:
                out = net(x)
            rows.copy_(l2_normalize_rows(out))
        return slab

    try:
        return fold_batches(run, slab, dataset, bs)
    finally:
        net.classifier = classifier


def get_class_net():
    net = TuneClassif(base_model(P), len(labels), untrained=P.untrained_blocks)
    return move_device(load_weights(net, P.preload_net), P.cuda_device)
