"""Dataset listing helpers (reference utils/dataset.py:11-22, 66-76) plus the in-memory
synthetic source the offline benchmarks use.  A dataset is a list of (tensor, label, path)
tuples, exactly as the reference's test mains build it (test/classif_finetune_test.py:62-73)."""
import glob
from os import path

import torch


def get_images_labels(folder='.', label_f=lambda x: x.split('.')[0]):
    """[(image filename, label)] for every jpg/JPG/JPEG/png of `folder`."""
    found = []
    for ext in ('*.jpg', '*.JPG', '*.JPEG', '*.png'):
        found.extend((im, label_f(im)) for im in glob.iglob(path.join(folder, ext)))
    return found


def get_lab_indicators(dataset, device):
    """label -> uint8 mask over the dataset marking the items carrying that label."""
    labs = [lab for _, lab, _ in dataset]
    out = {}
    for lab in labs:
        if lab not in out:
            mask = torch.tensor([1 if l2 == lab else 0 for l2 in labs], dtype=torch.uint8)
            out[lab] = mask.cuda() if device >= 0 else mask
    return out


def get_pos_couples(dataset, duplicate=True):
    """label -> [(label, (i1, i2), (x1, x2))] over all same-label index pairs i1 <= i2 (i1 < i2 without
    `duplicate`), in index order (reference utils/dataset.py:42-55)."""
    by_label = {}
    for i, (_, lab, _) in enumerate(dataset):
        by_label.setdefault(lab, []).append(i)
    couples = {}
    order = sorted((idx[0], lab) for lab, idx in by_label.items())
    for _, lab in order:
        idx = by_label[lab]
        out = []
        for a in range(len(idx)):
            for b in range(a if duplicate else a + 1, len(idx)):
                i1, i2 = idx[a], idx[b]
                out.append((lab, (i1, i2), (dataset[i1][0], dataset[i2][0])))
        if out:
            couples[lab] = out
    return couples


def choose_rand_neg(train_set, lab):
    """A random image whose label differs from `lab` (reference utils/dataset.py:59-63)."""
    import random
    while True:
        im, l2, _ = random.choice(train_set)
        if l2 != lab:
            return im


def choose_rand_neg_index(train_set, lab):
    """Index of a random image whose label differs from `lab`: choose_rand_neg resolved to an index, so that the random fall-back
    negatives of an epoch can be fixed when the epoch is created (identically on every data-parallel rank) instead of being drawn
    while the batches are built."""
    import random
    while True:
        k = random.randrange(len(train_set))
        if train_set[k][1] != lab:
            return k


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def synthetic_images(n, size=(3, 224, 224), seed=1234, device='cpu'):
    """SURVEY.md 8d images: U[0,1) fp32, seeded on the CPU, per-channel mean/std normalised."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, *size, generator=g)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return ((x - mean) / std).to(device)


def synthetic_image_set(n, n_labels, size=(3, 224, 224), seed=1234, prefix='synthetic', structure=0.0):
    """An in-memory dataset of (tensor, label, path) tuples; label = 'cNNN' by index mod n_labels.
    structure = 0: pure U[0,1) noise (labels carry no signal: retrieval at chance level).  structure = s in (0,1]:
    pixel = (1-s) * noise + s * pattern[label], one smooth random pattern per label (a 7x7 U[0,1) grid, bilinearly
    upsampled; pattern seed fixed so that query and gallery sets built with different `seed`s share the instances) --
    a retrieval task a network can actually solve, for the end-to-end parity tests."""
    if structure <= 0.0:
        imgs = synthetic_images(n, size, seed)
    else:
        g = torch.Generator().manual_seed(seed)
        noise = torch.rand(n, *size, generator=g)
        gp = torch.Generator().manual_seed(777)
        grid = torch.rand(n_labels, size[0], 7, 7, generator=gp)
        pat = torch.nn.functional.interpolate(grid, size=size[1:], mode='bilinear', align_corners=False)
        x = (1.0 - structure) * noise + structure * pat[torch.arange(n) % n_labels]
        mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
        std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
        imgs = (x - mean) / std
    return [(imgs[i], 'c%03d' % (i % n_labels), '%s/%06d.png' % (prefix, i)) for i in range(n)]


def synthetic_descriptors(N, M, D=2048, sigma=4.0, seed=0, device='cpu'):
    """SURVEY.md 8d descriptor set: L = N/10 centroids ~N(0,I); gallery row i has label i mod L and
    value centroid + sigma*N(0,I); queries likewise.  Rows are NOT yet normalised.
    Returns (Q, G, qlab, glab) with int32 labels."""
    g = torch.Generator().manual_seed(seed)
    L = max(1, N // 10)
    cent = torch.randn(L, D, generator=g)
    glab = torch.arange(N) % L
    qlab = torch.arange(M) % L
    G = cent[glab] + sigma * torch.randn(N, D, generator=g)
    Q = cent[qlab] + sigma * torch.randn(M, D, generator=g)
    return Q.to(device), G.to(device), qlab.int().to(device), glab.int().to(device)
