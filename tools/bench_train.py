#!/usr/bin/env python3
"""Side measurement of next-scope row f1 (BASELINE config 4) on ONE MI355X: siamese triplet training of
DescriptorNet(ResNet-50) with per-epoch hard-negative mining, reference hyper-parameters (batch 64 as 8 micro-batches
of 8, SGD lr 1e-3 momentum 0.9 wd 5e-4, BN frozen), synthetic 224x224 images.  Prints one JSON object.
    python tools/bench_train.py [--images 512] [--labels 64] [--epochs 2]
Multi-GPU data-parallel runs use the same entry point under torch.distributed.run (isx/dp.GradAllReducer)."""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--labels", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=2)
    args = ap.parse_args()
    from train import siamese_descriptor as sd
    from utils.dataset import get_pos_couples, synthetic_image_set
    torch.manual_seed(0); random.seed(0)
    P = sd.P
    P.cuda_device, P.cnn_model, P.feature_size2d, P.feature_dim = 0, "resnet50", (7, 7), 2048
    P.train_epochs, P.train_batch_size, P.train_micro_batch, P.test_batch_size = args.epochs, 64, 8, 128
    P.train_loss_int, P.train_test_int, P.untrained_blocks, P.train_epoch_switch = 10 ** 9, 10 ** 9, -1, 1
    tr = synthetic_image_set(args.images, args.labels, seed=1)
    te = synthetic_image_set(64, args.labels, seed=2)
    n_couples = sum(len(v) for v in get_pos_couples(tr).values())
    marks = []
    real = sd.mine_epoch_negatives

    def spy(*a, **k):                       # called once per epoch, before the training batches
        torch.cuda.synchronize()
        marks.append(time.perf_counter())
        return real(*a, **k)

    sd.mine_epoch_negatives = spy
    t0 = time.perf_counter()
    sd.main(tr, tr, te)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    marks.append(t1)
    per_epoch = [b - a for a, b in zip(marks[:-1], marks[1:])]
    steady = per_epoch[-1]
    print(json.dumps({"model": "DescriptorNet(resnet50, 2048)", "images": args.images, "labels": args.labels,
                      "triplets_per_epoch": n_couples, "epoch_seconds": per_epoch, "total_seconds": t1 - t0,
                      "triplets_per_s": n_couples / steady, "images_fwd_bwd_per_s": 3 * n_couples / steady,
                      "includes": "epoch embedding pass + isx_cosine_sim + isx_mine_negatives + forward/backward of 3 images per triplet + SGD"}))


if __name__ == "__main__":
    main()
