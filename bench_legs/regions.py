"""Side leg `extraction_regions` of bench.py.
side measurement: BASELINE configs[2] -- ResNet-50 region-pooled descriptors (classif_regions path) on 448 x 448 images,
then that config's retrieval leg at 1k queries x 100k gallery rows of those (class-score, 464-d) descriptors
"""
import json
import os
import sys
import time

from .common import (PEAK_F16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, RESNET50_GFLOP_PER_IMAGE, ROOT, build_net, load_traffic,
                     usable_cpus)


def measure(ctx):
    args, dev, ev, k, ops, rank, synthetic_descriptors, synthetic_images, torch = ctx.args, ctx.dev, ctx.ev, ctx.k, ctx.ops, ctx.rank, ctx.synthetic_descriptors, ctx.synthetic_images, ctx.torch
    from isx import backbones
    from model.nn_utils import fold_batch_norm, set_net_train
    from model.siamese import TuneClassifSub
    from train import classif_regions as cr
    Br, n_cls = args.regions_batch, 464
    torch.manual_seed(0)
    sub = TuneClassifSub(backbones.MODELS["resnet50"](pretrained=True, seed=0), n_cls, (7, 7))
    set_net_train(sub, False)
    sub.features = fold_batch_norm(sub.features)
    sub = sub.to(dev).to(memory_format=torch.channels_last)
    x_cpu = synthetic_images(8, size=(3, 448, 448), seed=4321 + rank)
    x = x_cpu.to(dev).repeat((Br + 7) // 8, 1, 1, 1)[:Br].contiguous(memory_format=torch.channels_last)
    slab = torch.empty((Br, n_cls), device=dev)

    def run():
        with torch.no_grad():
            slab.copy_(cr._best_location_descriptors(sub(x)[0]))       # features -> box pool -> 1x1 classifier -> best location -> L2 -> slab rows

    run(); run()
    torch.cuda.synchronize()
    n_it = 5
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(n_it):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms_ = e0.elapsed_time(e1) / n_it              # this rank's launches (no collective in here: a failure on one rank cannot hang the others)
    # which kernels ran (one instrumented launch): every convolution of the step must be a libisx entry point
    ops.KERNEL_TIMER = []
    run()
    torch.cuda.synchronize()
    timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    fams = {}
    for name, flop, nbytes, ea, eb in timer:
        f = fams.setdefault(name, {"launches": 0, "ms": 0.0, "flop": 0.0})
        f["launches"] += 1; f["ms"] += ea.elapsed_time(eb); f["flop"] += flop
    for f in fams.values():
        f["tflops"] = f["flop"] / (f["ms"] * 1e-3) / 1e12 if f["ms"] > 0 else None
    conv_flop = sum(f["flop"] for f in fams.values())
    alg_bytes = sum(nb for _, _, nb, _, _ in timer)           # algorithmic bytes of the convolutions of one launch (activations in + out + weights)
    tr_ = load_traffic().get("regions_leg") or {}
    traffic_r = tr_.get("bytes_per_launch") if tr_.get("images_per_launch") == Br else None      # PMC profile of THIS leg at this batch, else null
    flop_img = 4.0 * RESNET50_GFLOP_PER_IMAGE * 1e9 + 2.0 * 64 * 2048 * n_cls        # every convolution sees 4x the pixels of 224 x 224; + the 1x1 classifier on 8 x 8 locations
    ips = Br / (ms_ * 1e-3)
    assert bool(torch.isfinite(slab).all())
    res = {"workload": "BASELINE configs[2]: ResNet-50 TuneClassifSub (fp32, BN folded, NHWC) on 448x448 synthetic images -> 8x8 map of %d class scores "
                       "-> best-location descriptor (train/classif_regions.py:107-132), %d images per launch" % (n_cls, Br),
           "images_per_s": ips, "ms_per_launch": ms_, "images_per_launch": Br,
           "roofline": {"bound": "mfma", "achieved": flop_img * ips / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": flop_img * ips / 1e12 / PEAK_F32_MFMA_TFLOPS, "algorithmic_flop_per_image": flop_img, "traffic": traffic_r,
                        "traffic_unit": "HBM bytes per launch (every kernel of the leg)", "traffic_source": tr_.get("source") if traffic_r is not None else None,
                        "algorithmic_bytes_per_launch": alg_bytes,
                        "traffic_over_algorithmic": (traffic_r / alg_bytes) if traffic_r and alg_bytes else None},
           "libisx_convolution_flop_per_image": conv_flop / Br,
           "all_convolutions_in_libisx": bool(conv_flop / Br > 0.995 * flop_img),
           "kernel_families": fams}
    del sub, x
    torch.cuda.empty_cache()
    # retrieval leg of the same config: 1k queries x 100k gallery rows, exact scores + top-k + full-rank AP without a sort
    Mq, Nr = 1000, 100000
    Qc, Gc, ql, gl = synthetic_descriptors(Nr, Mq, n_cls, seed=7 + rank)
    Qd, Gd = ops.l2norm_rows(Qc.to(dev)), ops.l2norm_rows(Gc.to(dev))
    ql, gl = ql.to(dev), gl.to(dev)
    simr = torch.empty((Mq, Nr), device=dev)

    def leg(f, n=20):
        for _ in range(5):                          # sub-millisecond launches: warm the clocks up before timing
            f()
        torch.cuda.synchronize()
        a, b = ev(), ev(); a.record()
        for _ in range(n):
            f()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    t_sim = leg(lambda: ops.cosine_sim(Qd, Gd, out=simr))
    t_topk = leg(lambda: ops.topk_rows(simr, k))
    t_ap = leg(lambda: ops.average_precision_sim(simr, ql, gl))
    ap = ops.average_precision_sim(simr, ql, gl)
    res["retrieval_1000x100000"] = {"descriptor_dim": n_cls, "cosine_sim_ms": t_sim, "topk_rows_ms": t_topk, "average_precision_ms": t_ap,
                                    "total_ms": t_sim + t_topk + t_ap, "dist_per_s": Mq * Nr / ((t_sim + t_topk + t_ap) * 1e-3),
                                    "cosine_sim_tflops": 2.0 * Mq * Nr * n_cls / (t_sim * 1e-3) / 1e12,
                                    "cosine_sim_frac_of_f32_mfma_peak": 2.0 * Mq * Nr * n_cls / (t_sim * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                    "mAP": float(ap[~ap.isnan()].mean())}
    return res
