"""Global siamese descriptor approach: get_embeddings (reference
train/siamese_descriptor.py:25-41) and get_siamese_net (:151-162).  Triplet training with
hard-negative mining (:45-148) is the next scope item."""
import torch

from model.siamese import DescriptorNet, TuneClassif
from utils import fold_batches, move_device, tensor
from ._common import base_model, load_weights, stage_batch, test_transform
from .siamese_descriptor_p import P

labels = []


def get_embeddings(net, dataset, device, out_size):
    trans = test_transform(P)
    slab = tensor(device, len(dataset), out_size)

    def run(slab, i, is_final, batch):
        with torch.no_grad():
            slab[i:i + len(batch)].copy_(net(stage_batch(batch, trans, P.cuda_device)))
        return slab

    return fold_batches(run, slab, dataset, P.test_batch_size)


def get_siamese_net():
    class_net = TuneClassif(base_model(P), P.num_classes, untrained=P.untrained_blocks)
    load_weights(class_net, P.classif_model)
    net = DescriptorNet(class_net, P.feature_dim, P.feature_size2d, untrained=P.untrained_blocks)
    return move_device(load_weights(net, P.preload_net), P.cuda_device)
