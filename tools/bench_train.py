#!/usr/bin/env python3
"""Side measurement of next-scope row f1 (BASELINE config 4) on ONE MI355X: siamese triplet training of
DescriptorNet(ResNet-50) with per-epoch hard-negative mining, reference hyper-parameters (batch 64 as 8 micro-batches
of 8, SGD lr 1e-3 momentum 0.9 wd 5e-4, BN frozen), synthetic 224x224 images.  Two configurations:

  reference   untrained_blocks from the reference's table (train/siamese_descriptor_p.py:14-17,48; ResNet-50: 2+3+4+6 = 15):
              stem + layers 1-3 frozen (HIP trunk, no graph), layer4 + the descriptor head TRAINED
  frozen      untrained_blocks = -1: only the descriptor head learns (this repo's round-2/3 figure)

Prints one JSON object.
    python tools/bench_train.py [--images 512] [--labels 64] [--epochs 2] [--configs reference,frozen]
Data parallel: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py`
-- one rank per GPU (LOCAL_RANK), RCCL process group, every rank takes 1/N of each mini-batch's micro-batches (isx/dp.py),
weights broadcast from rank 0 by utils.train_gen, rank 0 prints the line."""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import torch  # noqa: E402


def run_config(name, args, world, rank, local):
    from train import siamese_descriptor as sd
    from utils.dataset import get_pos_couples, synthetic_image_set
    from isx import dp
    torch.manual_seed(0); random.seed(0)
    P = sd.P
    P.cuda_device, P.cnn_model, P.feature_size2d, P.feature_dim = local, args.backbone, (7, 7), 2048
    P.train_epochs, P.train_batch_size, P.train_micro_batch, P.test_batch_size = args.epochs, 64, 8, 128
    P.train_loss_int, P.train_test_int, P.train_epoch_switch = 10 ** 9, 10 ** 9, 1
    P.untrained_blocks = -1 if name == "frozen" else None          # None: the reference's table
    tr = synthetic_image_set(args.images, args.labels, seed=1)
    te = synthetic_image_set(64, args.labels, seed=2)
    n_couples = sum(len(v) for v in get_pos_couples(tr).values())
    n_steps = n_couples // P.train_batch_size
    marks = []
    real = sd.mine_epoch_negatives

    def spy(*a, **k):                       # called once per epoch, before the training batches
        torch.cuda.synchronize()
        marks.append(time.perf_counter())
        return real(*a, **k)

    sd.mine_epoch_negatives = spy
    dp.STATS.clear()
    import utils.train_general as tg
    tg.PHASES = {} if args.phases else None
    t0 = time.perf_counter()
    try:
        net, _ = sd.main(tr, tr, te)
    finally:
        sd.mine_epoch_negatives = real
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    marks.append(t1)
    per_epoch = [b - a for a, b in zip(marks[:-1], marks[1:])]
    steady = per_epoch[-1]
    trainable = sum(p.numel() for p in net.parameters() if p.requires_grad)
    out = {"untrained_blocks": P.untrained_blocks, "trainable_parameters": trainable,
           "trainable_modules": sorted(set(n.rsplit(".", 2)[0] if n.startswith("features.") else n.rsplit(".", 1)[0]
                                           for n, p in net.named_parameters() if p.requires_grad)),
           "triplets_per_epoch": n_couples, "optimizer_steps_per_epoch": n_steps, "epoch_seconds": per_epoch, "total_seconds": t1 - t0,
           "triplets_per_s": n_couples / steady, "images_fwd_bwd_per_s": 3 * n_couples / steady,
           "exchange": dict(dp.STATS)}
    if tg.PHASES:
        steps = n_steps * args.epochs
        out["phase_ms_per_step"] = dict((k, 1e3 * v / steps) for k, v in tg.PHASES.items())
        tg.PHASES = None
    del net
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512)
    ap.add_argument("--labels", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--backbone", default="resnet50")
    ap.add_argument("--configs", default="reference,frozen")
    ap.add_argument("--phases", action="store_true", help="synchronise and time the phases of the training step (diagnostic: the totals are slower)")
    args = ap.parse_args()
    import torch.distributed as dist
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    # debugging aid for boxes with ONE GPU (as in bench.py): ISX_BENCH_ONE_DEVICE=1 maps every rank to cuda:0 over gloo, so that the
    # data-parallel code path (micro-batch subtrees, TreeExchange, row all-gather) runs and its byte counts print; the rates mean nothing
    one_device = world > 1 and os.environ.get("ISX_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    res = {}
    for name in args.configs.split(","):
        res[name] = run_config(name, args, world, rank, local)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return
    line = {"model": "DescriptorNet(%s, 2048)" % args.backbone, "images": args.images, "labels": args.labels, "n_gpus": world,
            "collective_backend": (dist.get_backend() if False else ("gloo (all ranks on cuda:0)" if one_device else ("nccl" if world > 1 else None))),
            "includes": "epoch embedding pass + isx_cosine_sim + isx_mine_negatives + forward/backward of 3 images per triplet + SGD",
            "configs": res}
    if "reference" in res and "frozen" in res:
        line["reference_over_frozen"] = res["reference"]["triplets_per_s"] / res["frozen"]["triplets_per_s"]
    print(json.dumps(line))


if __name__ == "__main__":
    main()
