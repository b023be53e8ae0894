"""Region siamese descriptor approach: get_embeddings (reference
train/siamese_regions.py:26-41), get_siamese_net (:157-168) and the triplet training
(train_siam_triplets_pos_couples :45-154): triplet loss on the three region descriptors plus a
classification loss over the anchor's k selected windows, one triplet per micro-batch (the windows
differ per image), negatives mined for the whole epoch in one `isx_mine_negatives` launch."""
import torch
import torch.nn as nn

from model.siamese import RegionDescriptorNet, TuneClassifSub
from model.custom_modules import TripletLoss
from utils import (choose_rand_neg, choose_rand_neg_index, get_pos_couples, get_similarities, log, move_device, tensor,
                   test_print_descriptor, train_gen)
from ._common import base_model, device_batch_size, fold_shape_buckets, label_index, load_weights, make_resident, scatter_rows, test_transform
from .siamese_descriptor import mine_epoch_negatives, shuffle_couples
from .siamese_regions_p import P

labels = []


def get_embeddings(net, dataset, device, out_size):
    trans = test_transform(P)
    if trans is None:
        make_resident(dataset, P.cuda_device)
    slab = tensor(device, len(dataset), out_size)

    def run(indices, batch, x):
        with torch.no_grad():
            scatter_rows(slab, indices, net(x))

    # one image per step in the reference; images bucketed by shape share a backbone pass here, staged ahead of the trunk (BatchStager)
    fold_shape_buckets(run, dataset, lambda shape: device_batch_size(P, dataset, shape), stage=(trans, P.cuda_device))
    return slab


def get_siamese_net():
    class_net = TuneClassifSub(base_model(P), P.num_classes, P.feature_size2d, untrained=P.untrained_blocks)
    load_weights(class_net, P.classif_model)
    from model.siamese import head_weights_follow
    with head_weights_follow(bool(P.preload_net)):               # the state dict loaded below brings the head's weights
        net = RegionDescriptorNet(class_net, P.regions_k, P.feature_dim, P.feature_size2d, untrained=P.untrained_blocks)
    return move_device(load_weights(net, P.preload_net), P.cuda_device)


train_type = 'Siamese sub-regions'


def train_siam_triplets_pos_couples(net, train_set, testset_tuple, criterion, criterion2, optimizer, best_score=0):
    trans = None if P.train_pre_proc else P.train_trans
    couples = get_pos_couples(train_set)
    log(P, '#pos (without order, with duplicates):{0}'.format(sum(len(c) for c in couples.values())))

    def create_epoch(epoch, couples, testset_tuple):
        similarities, _ = get_similarities(P, get_embeddings, net, testset_tuple[1])
        shuffled = shuffle_couples(couples)
        negs = mine_epoch_negatives(similarities, testset_tuple[1], shuffled, epoch < P.train_epoch_switch).tolist()
        negs = [k if k >= 0 else choose_rand_neg_index(train_set, c[0]) for c, k in zip(shuffled, negs)]     # fixed per epoch, on every rank alike
        return [c + (k,) for c, k in zip(shuffled, negs)], {'epoch': epoch}

    def create_batch(batch, n, epoch):
        # one triplet at a time (reference :95-137): the selected windows differ per image
        lab, _, (im1, im2), k = batch[0]
        im3 = train_set[k][0] if k >= 0 else choose_rand_neg(train_set, lab)
        prep = (lambda im: im) if trans is None else trans
        mv = lambda im: move_device(prep(im).unsqueeze(0), P.cuda_device)
        return [mv(im1), mv(im2), mv(im3)], [move_device(torch.tensor([label_index(labels)[lab]], dtype=torch.int64), P.cuda_device)]

    def create_loss(out, labels_list):
        # triplet loss on the descriptors + classification loss over the anchor's k windows (:139-150)
        loss = criterion(*(d for d, _ in out))
        cls_all = out[0][1].squeeze(0).t()                               # (k, num_classes)
        loss2 = criterion2(cls_all, labels_list[0].expand(cls_all.size(0)))
        return loss, loss2

    return train_gen(train_type, P, test_print_descriptor, get_embeddings, net, couples, testset_tuple, optimizer, create_epoch,
                     create_batch, create_loss, best_score=best_score)


def main(train_set, test_train_set, test_set):
    """Training entry (reference :171-213) on already loaded (tensor, label, path) datasets."""
    from utils.train_general import make_sgd
    del labels[:]
    labels.extend(sorted(set(l for _, l, _ in train_set)))
    P.num_classes = len(labels)
    P.train_micro_batch = 1                                               # has to be 1 (reference siamese_regions_p.py:64)
    net = get_siamese_net()
    optimizer = make_sgd((p for p in net.parameters() if p.requires_grad), P.train_lr, P.train_momentum, P.train_weight_decay)
    criterion = TripletLoss(P.triplet_margin, P.train_loss_avg)
    criterion2 = nn.CrossEntropyLoss(reduction='mean' if P.train_loss2_avg else 'sum')
    testset_tuple = (test_set, test_train_set)
    score = test_print_descriptor(train_type, P, net, testset_tuple, get_embeddings) if getattr(P, 'test_upfront', True) else 0
    if not getattr(P, 'train', True):
        return net, score
    score = train_siam_triplets_pos_couples(net, train_set, testset_tuple, criterion, criterion2, optimizer, best_score=score)
    test_print_descriptor(train_type, P, net, testset_tuple, get_embeddings, best_score=len(test_set) + 1)
    return net, score


def run(dataset_full=None):
    """The reference's main() (:171-213): the sets come from P.dataset_full (a dataset folder with its `test` sub-folder, or a `synthetic:` spec),
    then upfront test (P.test_upfront) -> training (P.train) -> final test, as main() above does on loaded sets."""
    from ._common import load_training_sets
    return main(*load_training_sets(P, dataset_full or P.dataset_full, labels))


if __name__ == '__main__':
    import sys
    from ._common import training_cli
    training_cli(sys.argv[1:], P, run, 'train.siamese_regions')
