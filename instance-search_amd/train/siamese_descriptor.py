"""Global siamese descriptor approach: get_embeddings (reference
train/siamese_descriptor.py:25-41), get_siamese_net (:151-162) and the triplet training with
semi-hard / hard negative mining (train_siam_triplets_pos_couples :45-148).  Per epoch the whole
reference set is embedded, the N x N similarity matrix comes from `isx_cosine_sim`, and the negative
of EVERY positive couple of the epoch is mined in one `isx_mine_negatives` launch (the reference
walks the couples in Python, one masked row arg-max each)."""
import random

import numpy as np
import torch

from model.siamese import DescriptorNet, TuneClassif
from model.custom_modules import TripletLoss
from utils import (choose_rand_neg, choose_rand_neg_index, embeddings_device_dim, fold_batches, get_pos_couples, get_similarities, log, move_device,
                   tensor, test_print_descriptor, train_gen)
from ._common import BatchStager, base_model, device_batch_size, label_index, load_weights, make_resident, stage_batch, stage_images, test_transform
from .siamese_descriptor_p import P

labels = []


def get_embeddings(net, dataset, device, out_size):
    trans = test_transform(P)
    slab = tensor(device, len(dataset), out_size)
    if trans is None:
        make_resident(dataset, P.cuda_device)

    bs = device_batch_size(P, dataset)
    stager = BatchStager(dataset, bs, trans, P.cuda_device)    # resident: row gathers; beyond the HBM budget: double-buffered pinned staging on a copy stream

    def run(slab, i, is_final, batch):
        with torch.no_grad():
            slab[i:i + len(batch)].copy_(net(stager.get(i, batch)))
        return slab

    return fold_batches(run, slab, dataset, bs)


def get_siamese_net():
    class_net = TuneClassif(base_model(P), P.num_classes, untrained=P.untrained_blocks)
    load_weights(class_net, P.classif_model)
    from model.siamese import head_weights_follow
    with head_weights_follow(bool(P.preload_net)):               # the state dict loaded below brings the head's weights
        net = DescriptorNet(class_net, P.feature_dim, P.feature_size2d, untrained=P.untrained_blocks)
    return move_device(load_weights(net, P.preload_net), P.cuda_device)


def shuffle_couples(couples):
    """Interleave the couples so that every mini-batch sees many instances (reference :62-85): labels
    are shuffled, couples are dealt round-robin up to the 80th-percentile list length, the tail of the
    longer lists is inserted at random positions."""
    for lab in couples:
        random.shuffle(couples[lab])
    x = int(np.percentile(np.array([len(c) for c in couples.values()]), 80))
    keys = list(couples.keys())
    random.shuffle(keys)
    out = [couples[lab][i] for i in range(x) for lab in keys if i < len(couples[lab])]
    for lab in keys:
        for c in couples[lab][x:]:
            out.insert(random.randrange(len(out)), c)
    return out


def mine_epoch_negatives(similarities, dataset, couples_list, semi_hard):
    """Index of the negative for every couple (-1: none, use a random one)."""
    table = {}
    lab = torch.tensor([table.setdefault(l, len(table)) for _, l, _ in dataset], dtype=torch.int32)
    i1 = torch.tensor([c[1][0] for c in couples_list], dtype=torch.int64)
    i2 = torch.tensor([c[1][1] for c in couples_list], dtype=torch.int64)
    if not isinstance(similarities, torch.Tensor):
        # SimilarityRows: the n x n matrix is over budget -- mine block by block of anchor rows (couples grouped by the block
        # their anchor i1 falls in)
        from utils.metrics import row_blocks
        n = similarities.size(0)
        neg = torch.full_like(i1, -1)
        for r0, r1 in row_blocks(n, n):
            sel = ((i1 >= r0) & (i1 < r1)).nonzero().flatten()
            if sel.numel():
                neg[sel] = _mine_block(similarities.rows(r0, r1), lab, i1[sel], i2[sel], semi_hard, r0)
        return neg
    return _mine_block(similarities, lab, i1, i2, semi_hard, 0)


def _mine_block(similarities, lab, i1, i2, semi_hard, row_base):
    """similarities: rows [row_base, row_base + rows) of the n x n matrix; i1 (anchors, all inside the block) and i2 absolute."""
    if similarities.is_cuda:
        from isx import ops
        return ops.mine_negatives(similarities, lab.cuda(), i1.cuda(), i2.cuda(), semi_hard, row_base=row_base).cpu()
    anchor_lab = lab[i1]
    i1 = i1 - row_base
    rows = similarities[i1]
    excl = lab[None, :] == anchor_lab[:, None]
    if semi_hard:
        excl = excl | (rows >= similarities[i1, i2][:, None])
    masked = rows.masked_fill(excl, -2.0)
    neg = masked.argmax(1)
    return torch.where(excl.all(1), torch.full_like(neg, -1), neg)


def train_siam_triplets_pos_couples(net, train_set, testset_tuple, criterion, optimizer, best_score=0):
    trans = None if P.train_pre_proc else P.train_trans
    if trans is None:
        make_resident(train_set, P.cuda_device)          # batches become row gathers on the device
    couples = get_pos_couples(train_set)
    log(P, '#pos (without order, with duplicates):{0}'.format(sum(len(c) for c in couples.values())))

    def create_epoch(epoch, couples, testset_tuple):
        similarities, _ = get_similarities(P, get_embeddings, net, testset_tuple[1])
        shuffled = shuffle_couples(couples)
        negs = mine_epoch_negatives(similarities, testset_tuple[1], shuffled, epoch < P.train_epoch_switch).tolist()
        missing = sum(1 for k in negs if k < 0)
        if missing:
            log(P, 'cant find semi-hard neg for {0} couples, falling back to random neg'.format(missing))
        # the random fall-back (reference :108-131 draws it while the batch is built) is drawn here, once per epoch and in couple order: every
        # data-parallel rank holds the same list, whatever slice of the mini-batches it will run
        negs = [k if k >= 0 else choose_rand_neg_index(train_set, c[0]) for c, k in zip(shuffled, negs)]
        tagged = [c + (k,) for c, k in zip(shuffled, negs)]
        return tagged, {'epoch': epoch}

    def create_batch(batch, n, epoch):
        prep = (lambda im: im) if trans is None else trans
        a = stage_images([prep(im1) for _, _, (im1, _), _ in batch], P.cuda_device)
        p = stage_images([prep(im2) for _, _, (_, im2), _ in batch], P.cuda_device)
        ng = stage_images([prep(train_set[k][0] if k >= 0 else choose_rand_neg(train_set, lab)) for lab, _, _, k in batch], P.cuda_device)
        ids = label_index(labels)
        lab_ids = torch.tensor([ids[lab] for lab, _, _, _ in batch], dtype=torch.int64)
        return [a, p, ng], [move_device(lab_ids, P.cuda_device)]

    # same items -> same batch, whenever it is built: the negatives were drawn in create_epoch and nothing is augmented, so the training step may
    # build the batches of the next mini-batches ahead of their turn (utils/train_general._Stepper._precompute_ahead)
    create_batch.deterministic = trans is None

    def create_loss(out, labels_list):
        return criterion(*out), None

    # the loss IS the triplet criterion on the net's three outputs: the step may evaluate it for all its micro-batches in one launch
    # (utils/train_general._Stepper._leaves_batched -> isx_triplet_leaves), same values per row
    from model.custom_modules import TripletLoss
    if type(criterion) is TripletLoss:
        create_loss.triplet = criterion

    return train_gen(train_type, P, test_print_descriptor, get_embeddings, net, couples, testset_tuple, optimizer, create_epoch,
                     create_batch, create_loss, best_score=best_score)


train_type = 'Siamese descriptor'


def main(train_set, test_train_set, test_set):
    """Training entry (reference :165-207) on already loaded (tensor, label, path) datasets."""
    from utils.train_general import make_sgd
    del labels[:]
    labels.extend(sorted(set(l for _, l, _ in train_set)))
    P.num_classes = len(labels)
    net = get_siamese_net()
    optimizer = make_sgd((p for p in net.parameters() if p.requires_grad), P.train_lr, P.train_momentum, P.train_weight_decay)
    criterion = TripletLoss(P.triplet_margin, P.train_loss_avg)
    testset_tuple = (test_set, test_train_set)
    score = test_print_descriptor(train_type, P, net, testset_tuple, get_embeddings) if getattr(P, 'test_upfront', True) else 0
    if not getattr(P, 'train', True):
        return net, score
    score = train_siam_triplets_pos_couples(net, train_set, testset_tuple, criterion, optimizer, best_score=score)
    test_print_descriptor(train_type, P, net, testset_tuple, get_embeddings, best_score=len(test_set) + 1)
    return net, score


def run(dataset_full=None):
    """The reference's main() (:165-207): the sets come from P.dataset_full (a dataset folder with its `test` sub-folder, or a `synthetic:` spec),
    then upfront test (P.test_upfront) -> training (P.train) -> final test, as main() above does on loaded sets."""
    from ._common import load_training_sets
    return main(*load_training_sets(P, dataset_full or P.dataset_full, labels))


if __name__ == '__main__':
    import sys
    from ._common import training_cli
    training_cli(sys.argv[1:], P, run, 'train.siamese_descriptor')
