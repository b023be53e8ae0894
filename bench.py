#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on N MI355X GPUs of one node.

Workload (BASELINE configs[1], "ResNet-50 fully-conv global descriptors, 10k-image synthetic
gallery"): one STEP = one batch of synthetic 224x224 images through the whole path
    images --ResNet-50 convs (PyTorch-ROCm / MIOpen)--> (B,2048,7,7) feature map
           --isx_gap_l2 (HIP)--> L2-normalised 2048-d descriptors
           --[N>1: RCCL all-gather of the query descriptors]--
           --isx_cosine_sim (fp32 MFMA) + isx_topk_rows (HIP) against this rank's gallery shard-->
           --[N>1: RCCL all-gather of per-shard top-k + isx_topk_merge]--> ranked top-100 lists
Inputs (images, gallery slab) are resident in HBM before the timed region.  Weak scaling: every
rank extracts B images and holds a 10k-row gallery shard.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 3

Rank 0 prints ONE JSON line (metric / value / roofline / cpu_baseline ...).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F16_MFMA_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak of one MI355X (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0            # HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU per step")
    ap.add_argument("--gallery", type=int, default=10000, help="gallery rows per GPU")
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--backbone", default="resnet50")
    ap.add_argument("--backbone-dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = the reference's precision (default); bf16 is reported separately, never as `value`")
    ap.add_argument("--memory-format", default="channels_last", choices=["channels_last", "contiguous"],
                    help="layout of the backbone activations (same fp32 math; MIOpen's NHWC kernels are ~9%% faster); "
                         "isx_gap_l2 consumes either layout in place")
    ap.add_argument("--no-fold-bn", action="store_true",
                    help="keep BatchNorm as separate kernels (default: folded into the convolutions for inference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shard-bench", action="store_true", help="skip the 10k x 125k retrieval-shard side measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def build_net(name, dtype, device, channels_last=False, fold_bn=False):
    from isx import backbones
    from model.nn_utils import set_net_train
    from model.siamese import TuneClassif
    torch.manual_seed(0)
    net = TuneClassif(backbones.MODELS[name](pretrained=True, seed=0), 464)
    set_net_train(net, False)
    if fold_bn:
        from model.nn_utils import fold_batch_norm
        net.features = fold_batch_norm(net.features)
    net = net.to(device)
    if dtype == "bf16" or channels_last:
        net = net.to(memory_format=torch.channels_last)
    return net


def usable_cpus():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(args, gallery_cpu, images_cpu):
    """The reference's CPU path (torch CPU backbone, then the oracle's pooling / cosine / top-k)
    on a bounded sample of the same workload; images/s on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    net = build_net(args.backbone, "f32", "cpu")
    threads = min(usable_cpus(), 32)           # torch's conv scaling flattens out beyond a few dozen threads at this batch
    torch.set_num_threads(threads)
    G = gallery_cpu.numpy()

    def run(n):
        x = images_cpu[:n]
        with torch.no_grad():
            fmap = net.features(x)
        q = O.gap_l2(fmap.numpy())
        O.cosine_topk(q, G, args.k)

    run(2)                                                  # warm caches / thread pool
    t0 = time.time(); run(4); per = (time.time() - t0) / 4
    nb = images_cpu.size(0)
    passes = int(max(1, min(16, round((args.cpu_seconds - 6 * per) / max(per * nb, 1e-3)))))   # ~cpu_seconds of work
    t0 = time.time()
    for _ in range(passes):
        run(nb)
    dt = time.time() - t0
    n = passes * nb
    return {"value": n / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "%d images: torch-CPU fp32 %s features + oracle gap_l2 + oracle cosine_topk vs the %d-row gallery, %.1f s"
                      % (n, args.backbone, G.shape[0], dt)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the HIP path)")
    # debugging aid for boxes with ONE GPU: ISX_BENCH_ONE_DEVICE=1 maps every rank to cuda:0 and uses gloo, so the N > 1 code path
    # (query all-gather, per-shard search, result all-gather, merge) can be exercised; the numbers it prints mean nothing
    one_device = world > 1 and os.environ.get("ISX_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and one_device:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from isx import ops, retrieval
    from utils.dataset import synthetic_descriptors, synthetic_images

    B, Ng, D, k = args.batch, args.gallery, 2048, args.k
    # ---- resident inputs -------------------------------------------------------------
    images_cpu = synthetic_images(min(B, 64), seed=1234 + rank)
    images = images_cpu.to(dev).repeat((B + images_cpu.size(0) - 1) // images_cpu.size(0), 1, 1, 1)[:B].contiguous()
    cl = args.memory_format == "channels_last" or args.backbone_dtype == "bf16"
    if cl:
        images = images.to(memory_format=torch.channels_last)
    _, G_cpu, _, _ = synthetic_descriptors(Ng, 1, D, seed=rank)
    shard = ops.l2norm_rows(G_cpu.to(dev))
    gallery = retrieval.ShardedGallery(shard, idx_base=rank * Ng)
    net = build_net(args.backbone, args.backbone_dtype, dev, channels_last=cl, fold_bn=not args.no_fold_bn)
    q_local = torch.empty((B, D), device=dev)
    M = B * world
    sim = torch.empty((M, Ng), device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    gemm_ev, gap_ev = [], []

    def step(timed):
        if timed:
            a, b = ev(), ev()
        with torch.no_grad():
            if args.backbone_dtype == "bf16":
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    fmap = net.features(images)
                fmap = fmap.float()
            else:
                fmap = net.features(images)
        if timed:
            a.record()
        ops.gap_l2(fmap, out=q_local)
        if timed:
            b.record(); gap_ev.append((a, b))
        Q = retrieval.gather_queries(q_local)
        if timed:
            c, d = ev(), ev(); c.record()
        ops.cosine_sim(Q, shard, out=sim)
        if timed:
            d.record(); gemm_ev.append((c, d))
        s, i = ops.topk_rows(sim, k, idx_base=gallery.idx_base)
        if world > 1:
            all_s = torch.empty((world, M, k), dtype=s.dtype, device=dev)
            all_i = torch.empty((world, M, k), dtype=i.dtype, device=dev)
            dist.all_gather_into_tensor(all_s.view(-1, k), s)
            dist.all_gather_into_tensor(all_i.view(-1, k), i)
            s, i = ops.topk_merge(all_s, all_i)
        return s, i

    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops.KERNEL_TIMER = []                     # per-launch HIP events around the hand-written trunk convolutions
    for _ in range(args.steps):
        out = step(True)
    torch.cuda.synchronize()
    trunk_timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER, None
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert out[1].shape == (M, k) and int(out[1].min()) >= 0

    gemm_ms = sum(a.elapsed_time(b) for a, b in gemm_ev) / len(gemm_ev)
    gap_ms = sum(a.elapsed_time(b) for a, b in gap_ev) / len(gap_ev)
    trunk = {}
    for name, flop, nbytes, ea, eb in trunk_timer:
        t = trunk.setdefault(name, [0, 0.0, 0.0, 0.0])
        t[0] += 1; t[1] += flop; t[2] += nbytes; t[3] += ea.elapsed_time(eb)
    gemm_flop = 2.0 * M * Ng * D
    gap_bytes = B * D * 49 * 4 + B * D * 4

    # side measurement: BASELINE configs[4] -- 10k replicated queries against a gallery sharded 125k rows per
    # GPU (1M rows at 8 GPUs): local fused top-k + all-gather of the per-shard lists + merge, end to end
    shard_result = None

    def shard_bench():
        Ms, Ns = 10000, 125000
        gq = torch.Generator(device=dev).manual_seed(1)
        Qs = ops.l2norm_rows(torch.randn(Ms, D, device=dev, generator=gq))              # same queries on every rank
        gg = torch.Generator(device=dev).manual_seed(100 + rank)
        Gs = ops.l2norm_rows(torch.randn(Ns, D, device=dev, generator=gg))

        def time_search(fast):
            gal = retrieval.ShardedGallery(Gs, idx_base=rank * Ns, fast=fast)
            gal.search(Qs, k)                      # warm-up (fast: builds the cached fp16 image of the shard)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            ts0 = time.perf_counter()
            for _ in range(3):
                res = gal.search(Qs, k)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            ms_ = (time.perf_counter() - ts0) / 3 * 1e3
            if world > 1:
                tm_ = torch.tensor([ms_], device=dev, dtype=torch.float64)
                dist.all_reduce(tm_, op=dist.ReduceOp.MAX)
                ms_ = float(tm_.item())
            return ms_, res

        ms32, (rs32, ri32) = time_search(False)      # every score on the fp32 matrix cores
        ms, (rs, ri) = time_search(True)             # fp16-MFMA filter + exact fp32 re-scoring: must be identical
        identical = bool(torch.equal(ri, ri32) and torch.equal(rs.view(torch.int32), rs32.view(torch.int32)))
        if not identical:                                # reported in the JSON line; never silently, never fatal for the headline number
            print("bench.py: WARNING isx_cosine_topk_fast differs from isx_cosine_topk on the shard workload", file=sys.stderr)
        assert ri.shape == (Ms, k) and int(ri.min()) >= 0 and int(ri.max()) < Ns * world
        flop = 2.0 * Ms * Ns * world * D
        return {"shape": [Ms, Ns * world, D], "gallery_rows_per_gpu": Ns, "k": k, "ms": ms,
                        "dist_per_s": Ms * Ns * world / (ms * 1e-3),
                        "tflops_end_to_end": flop / (ms * 1e-3) / 1e12,
                        "frac_of_f16_mfma_peak": flop / (ms * 1e-3) / 1e12 / (PEAK_F16_MFMA_TFLOPS * world),
                        "path": "isx_cosine_topk_fast (fp16-MFMA filter + exact fp32 re-scoring, bit-identical results)",
                        "fp32_path": {"ms": ms32, "dist_per_s": Ms * Ns * world / (ms32 * 1e-3),
                                      "tflops_end_to_end": flop / (ms32 * 1e-3) / 1e12,
                                      "frac_of_f32_mfma_peak": flop / (ms32 * 1e-3) / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)},
                        "identical_to_fp32_path": identical,
                        "includes": "local top-k" + (" + RCCL all-gather of per-shard top-k + isx_topk_merge" if world > 1 else "")}


    if not args.no_shard_bench:
        if world > 1:
            shard_result = shard_bench()             # collective inside: every rank must take the same path
        else:
            try:
                shard_result = shard_bench()
            except Exception as e:                   # a failed side measurement must not cost the headline number
                shard_result = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    if rank == 0:
        images_per_s = world * B * args.steps / dt
        traffic = None
        prof = os.path.join(ROOT, "profiles", "roofline_traffic.json")
        if os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get("cosine_gemm_kernel", {}).get("%dx%dx%d" % (M, Ng, D))
            except Exception:
                traffic = None
        line = {
            "metric": "images/sec descriptor extract + query x gallery search",
            "value": images_per_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ResNet-50 fully-conv global descriptors + top-%d cosine search, "
                                   "%d-row gallery shard per GPU, 224x224 synthetic images" % (k, Ng),
                       "images_per_gpu_per_step": B, "gallery_rows_per_gpu": Ng, "descriptor_dim": D, "k": k,
                       "backbone": args.backbone, "backbone_dtype": args.backbone_dtype, "activation_layout": "NHWC" if cl else "NCHW", "bn_folded": not args.no_fold_bn, "parallelism": "gallery-row shards x%d + DP extraction" % world},
            "dist_per_s": images_per_s * Ng * world,
            "roofline": {"kernel": "cosine_gemm_kernel (isx_cosine_sim, v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                         "achieved": gemm_flop / (gemm_ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": gemm_flop / (gemm_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                         "launch_ms": gemm_ms, "algorithmic_flop_per_launch": gemm_flop, "shape": [M, Ng, D]},
            "roofline_gap_l2": {"kernel": "gap_l2_nhwc_kernel (isx_gap_l2_nhwc)" if cl else "gap_l2_kernel (isx_gap_l2)", "bound": "hbm", "achieved": gap_bytes / (gap_ms * 1e-3) / 1e9,
                                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gap_bytes / (gap_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                "launch_ms": gap_ms, "algorithmic_bytes_per_launch": gap_bytes},
        }
        for name, (cnt, flop, nbytes, ms_) in sorted(trunk.items()):
            # all launches of one hand-written trunk kernel over the timed steps: algorithmic FLOP (and bytes) / summed HIP-event time
            line["roofline_" + name] = {"kernel": name + (" = cosine_gemm_kernel<EPI=2>" if "1x1" in name else " = conv3x3_nhwc_kernel") +
                                        " (v_mfma_f32_32x32x2_f32, bias/residual/ReLU fused)", "bound": "mfma",
                                        "achieved": flop / (ms_ * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                        "frac": flop / (ms_ * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "launches_per_step": cnt // args.steps,
                                        "ms_per_step": ms_ / args.steps, "algorithmic_GBps": nbytes / (ms_ * 1e-3) / 1e9}
        if shard_result is not None:
            line["retrieval_shard"] = shard_result
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args, shard.cpu(), images_cpu)
            except Exception as e:
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
