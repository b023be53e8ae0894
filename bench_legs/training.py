"""Side leg `training` of bench.py: next-scope row f1 (BASELINE configs[3] on ONE GPU) -- siamese triplet training of DescriptorNet(ResNet-50) on the
reference's configuration (layer4 + head trained), with the whole trunk frozen, and with the prefix-feature cache: tools/bench_train.py (N = 1 only)."""
import os
import sys

from .common import ROOT


def measure(ctx):
    local = ctx.local
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_train
    import io
    import contextlib
    targs = bench_train.make_parser().parse_args(["--images", "512", "--labels", "64", "--epochs", "5", "--backbone", "resnet50"])   # the tool's own defaults
    res_t = {}
    from train import siamese_descriptor as _sd
    saved_p = dict(_sd.P.__dict__)
    try:
        with contextlib.redirect_stdout(io.StringIO()):          # the training script logs its evaluation lines to stdout
            for name in ("reference", "frozen", "reference_cached"):
                res_t[name] = bench_train.run_config(name, targs, 1, 0, local)
    finally:
        _sd.P.__dict__.clear(); _sd.P.__dict__.update(saved_p)
    return {"workload": "BASELINE configs[3] on one GPU: DescriptorNet(ResNet-50, 2048) triplet training with per-epoch hard-negative mining, "
                        "batch 64 = 8 micro-batches of 8, SGD 1e-3 / 0.9 / 5e-4, BN frozen, 512 synthetic images / 64 labels, 36 steps per epoch, 5 epochs",
            "reference_config_triplets_per_s": res_t["reference"]["triplets_per_s"],
            "reference_config": "untrained_blocks = 15 (reference train/siamese_descriptor_p.py:14-17,48): layer4 + descriptor head trained",
            "frozen_trunk_triplets_per_s": res_t["frozen"]["triplets_per_s"],
            "reference_over_frozen": res_t["reference"]["triplets_per_s"] / res_t["frozen"]["triplets_per_s"],
            "reference_config_with_prefix_cache_triplets_per_s": res_t["reference_cached"]["triplets_per_s"],
            "prefix_cache": "P.train_prefix_cache (off in the two figures above): frozen-prefix features of the resident training images looked up in an "
                            "HBM table instead of recomputed at every use; bit-identical training, not the reference's work per step",
            "prefix_look_ahead": "P.train_prefix_ahead = %d: the frozen prefix of that many consecutive mini-batches runs as one launch (every image still "
                                 "computed at every use; bit-identical to a launch per step)" % res_t["reference"].get("prefix_ahead", 1),
            "statistic": res_t["reference"]["statistic"],
            "reference_config_triplets_per_s_min_max": res_t["reference"]["triplets_per_s_min_max"],
            "frozen_trunk_triplets_per_s_min_max": res_t["frozen"]["triplets_per_s_min_max"],
            "roofline": dict((k, res_t["reference"]["roofline"][k]) for k in ("bound", "achieved", "peak", "unit", "frac", "ms_per_step",
                                                                               "algorithmic_flop_per_step", "phases_flop", "counts")),
            "trainable_parameters": res_t["reference"]["trainable_parameters"],
            "epoch_seconds": res_t["reference"]["epoch_seconds"], "exchange": res_t["reference"]["exchange"]}
