#!/usr/bin/env python3
"""Side measurements of the other BASELINE configs on one MI355X (not the driver's bench line):
extraction rate of the four approaches, retrieval + metric rates at config 2/3 scale.
Prints one JSON object.   python tools/bench_paths.py [--quick]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import torch  # noqa: E402

from isx import backbones, ops  # noqa: E402
from model.siamese import DescriptorNet, RegionDescriptorNet, TuneClassif, TuneClassifSub  # noqa: E402
from train import classif_regions as cr  # noqa: E402
from utils.dataset import synthetic_descriptors  # noqa: E402


def timed(f, n=3, w=1):
    """Seconds per call.  Calls shorter than a few milliseconds are repeated until the timed region holds >= 40 ms of work (and the
    warm-up as much): three launches of a sub-millisecond kernel out of an idle chip measure its clock ramp, not the kernel."""
    for _ in range(w):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    f()
    torch.cuda.synchronize()
    one = time.perf_counter() - t
    if one < 0.013:
        n = max(n, min(400, int(0.04 / max(one, 1e-5))))
        for _ in range(n):
            f()
        torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    quick = "--quick" in sys.argv
    out = {}
    dev = "cuda"
    with torch.no_grad():
        # ---- extraction (fp32, channels-last activations, synthetic images)
        x224 = torch.randn(256, 3, 224, 224, device=dev).to(memory_format=torch.channels_last)
        x448 = torch.randn(128, 3, 448, 448, device=dev).to(memory_format=torch.channels_last)
        from model.nn_utils import fold_batch_norm

        def cl(m):
            m = m.eval()
            if any(isinstance(x, torch.nn.BatchNorm2d) for x in m.features.modules()):
                m.features = fold_batch_norm(m.features)        # inference trunk: BN folded, fused epilogues
            return m.to(dev).to(memory_format=torch.channels_last)
        g = cl(TuneClassif(backbones.resnet50(pretrained=True), 464))
        slab = torch.empty(256, 2048, device=dev)
        t = timed(lambda: ops.gap_l2(g.features(x224), out=slab))
        out["extract_global_resnet50_224"] = {"images_per_s": 256 / t}
        # the reference's own ResNet (utils/general.py:39-44 admits alexnet | resnet152): 23.1 GFLOP of convolutions per 224 x 224 image
        del g
        x512 = torch.randn(512, 3, 224, 224, device=dev).to(memory_format=torch.channels_last)
        g152 = cl(TuneClassif(backbones.resnet152(pretrained=True), 464))
        slab512 = torch.empty(512, 2048, device=dev)
        t = timed(lambda: ops.gap_l2(g152.features(x512), out=slab512))
        out["extract_global_resnet152_224"] = {"images_per_s": 512 / t, "images_per_launch": 512, "gflop_per_image": 23.1,
                                               "frac_of_f32_mfma_peak": 23.1e9 * 512 / t / 157.3e12}
        del g152, x512, slab512
        sub = cl(TuneClassifSub(backbones.resnet50(pretrained=True), 464, (7, 7)))
        t = timed(lambda: cr._best_location_descriptors(sub(x448)[0]))
        out["extract_classif_regions_resnet50_448"] = {"images_per_s": 128 / t, "images_per_launch": 128, "map": "8x8 locations, 464 classes",
                                                       "frac_of_f32_mfma_peak": 32.8e9 * 128 / t / 157.3e12}
        dn = cl(DescriptorNet(backbones.resnet50(pretrained=True), 2048, (7, 7)))
        t = timed(lambda: dn(x224))
        out["extract_siamese_descriptor_resnet50_224"] = {"images_per_s": 256 / t, "head": "Linear(100352->2048)"}
        rd = cl(RegionDescriptorNet(backbones.resnet50(pretrained=True), 6, 2048, (7, 7)))
        t = timed(lambda: rd(x448))
        out["extract_siamese_regions_resnet50_448"] = {"images_per_s": 128 / t, "images_per_launch": 128, "k": 6}
        a = cl(TuneClassif(backbones.alexnet(pretrained=True), 464))
        a.classifier = torch.nn.Sequential()
        t = timed(lambda: ops.l2norm_rows(a(x224)))
        out["extract_global_alexnet_224"] = {"images_per_s": 256 / t}
        # BASELINE configs[0]'s descriptor on the GPU (extension): AlexNet fc7 = classifier[:6], 4096-d
        from train.classif_finetune import fc7_tap
        a7 = cl(TuneClassif(backbones.alexnet(pretrained=True), 464))
        a7.classifier = fc7_tap(a7.classifier)
        t = timed(lambda: ops.l2norm_rows(a7(x224)))
        out["extract_fc7_alexnet_224"] = {"images_per_s": 256 / t, "descriptor_dim": 4096}
        del a7
        del sub, dn, rd, a, x224, x448
        torch.cuda.empty_cache()
        # ---- retrieval + metrics
        for (M, N) in ((1000, 10000),) + (() if quick else ((1000, 100000),)):
            Q, G, ql, gl = synthetic_descriptors(N, M, 2048, device=dev)
            Q, G = ops.l2norm_rows(Q), ops.l2norm_rows(G)
            sim = torch.empty(M, N, device=dev)
            t_sim = timed(lambda: ops.cosine_sim(Q, G, out=sim))
            t_ap = timed(lambda: ops.average_precision_sim(sim, ql, gl))
            t_p1 = timed(lambda: ops.topk_rows(sim, 1))
            t_rank = timed(lambda: ops.average_precision(ops.rank_full(sim), ql, gl), n=2)
            ap = ops.average_precision_sim(sim, ql, gl).cpu()
            # top-k only (no full score matrix): the exact search with the fp16-MFMA filter vs scores + top-k
            from isx import retrieval
            gal = retrieval.ShardedGallery(G, 0)
            gal.search(Q, 100)
            t_fast = timed(lambda: gal.search(Q, 100))
            fs, fi = gal.search(Q, 100)
            ps, pi = ops.topk_rows(sim, 100)
            same = bool(torch.equal(fi, pi) and torch.equal(fs.view(torch.int32), ps.view(torch.int32)))
            t_top = timed(lambda: ops.topk_rows(sim, 100))
            out["search_top100_%dx%d" % (M, N)] = {"cosine_topk_fast_ms": t_fast * 1e3, "cosine_sim_plus_topk_rows_ms": (t_sim + t_top) * 1e3,
                                                    "identical": same}
            out["retrieval_%dx%d" % (M, N)] = {
                "cosine_sim_ms": t_sim * 1e3, "tflops": 2.0 * M * N * 2048 / t_sim / 1e12, "dist_per_s": M * N / t_sim,
                "ap_sort_free_ms": t_ap * 1e3, "p_at_1_ms": t_p1 * 1e3, "rank_full_plus_ap_ms": t_rank * 1e3,
                "mAP": float(ap[~ap.isnan()].mean())}
    # DBA (test/instance_avg.py) at gallery scale: 100 000 descriptors, 10 000 instances -- no N x N matrix (isx_dba_groups)
    if not quick:
        from test.instance_avg import instance_avg
        N, L = 100000, 10000
        E = ops.l2norm_rows(torch.randn(N, 2048, device=dev))
        ds = [(None, i * 7919 % L, None) for i in range(N)]
        instance_avg(0, E, ds, None, -1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        instance_avg(0, E, ds, None, -1)
        torch.cuda.synchronize()
        out["dba_100000x2048_10000_instances"] = {"seconds": time.perf_counter() - t0, "includes": "label grouping on the host + isx_dba_groups"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
