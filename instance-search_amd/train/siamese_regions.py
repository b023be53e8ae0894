"""Region siamese descriptor approach: get_embeddings (reference
train/siamese_regions.py:26-41) and get_siamese_net (:157-168)."""
import torch

from model.siamese import RegionDescriptorNet, TuneClassifSub
from utils import fold_batches, move_device, tensor
from ._common import base_model, load_weights, stage_batch, test_transform
from .siamese_regions_p import P

labels = []


def get_embeddings(net, dataset, device, out_size):
    trans = test_transform(P)
    slab = tensor(device, len(dataset), out_size)

    def run(slab, i, is_final, batch):
        with torch.no_grad():
            slab[i:i + len(batch)].copy_(net(stage_batch(batch, trans, P.cuda_device)))
        return slab

    # one image per step in the reference; same-sized images may share a backbone pass
    same = len(set(tuple(im.shape) for im, _, _ in dataset)) <= 1
    return fold_batches(run, slab, dataset, max(P.test_batch_size, 1) if same else 1)


def get_siamese_net():
    class_net = TuneClassifSub(base_model(P), P.num_classes, P.feature_size2d, untrained=P.untrained_blocks)
    load_weights(class_net, P.classif_model)
    net = RegionDescriptorNet(class_net, P.regions_k, P.feature_dim, P.feature_size2d, untrained=P.untrained_blocks)
    return move_device(load_weights(net, P.preload_net), P.cuda_device)
