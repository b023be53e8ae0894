"""The rank-0 record of a bench.py run: rooflines per kernel family from the instrumented pass, the driver's `roofline`, the side legs."""
from .common import (PEAK_F16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, RESNET50_GFLOP_PER_IMAGE, csrc_digest, load_traffic)
from .line import _pick


def assemble(ctx):
    """Rank 0: the measurements of the run -> the full record (contract fields, one roofline object per kernel family with its per-layer-shape table,
    `roofline` = the dominant family + north_star's named kernels, the side legs).  cpu_baseline is added by the caller."""
    B, D, M, Ng, args, backend, cl, dt, exchange_legs, gap_bytes, gap_ms, gemm_flop, gemm_ms, ingest_decode_result, ingest_result, k, ksteps, merge_identical, overlap, overlap_identical, regions_result, retrieval, shard_result, slab_result, training_result, trunk, world = ctx.B, ctx.D, ctx.M, ctx.Ng, ctx.args, ctx.backend, ctx.cl, ctx.dt, ctx.exchange_legs, ctx.gap_bytes, ctx.gap_ms, ctx.gemm_flop, ctx.gemm_ms, ctx.ingest_decode_result, ctx.ingest_result, ctx.k, ctx.ksteps, ctx.merge_identical, ctx.overlap, ctx.overlap_identical, ctx.regions_result, ctx.retrieval, ctx.shard_result, ctx.slab_result, ctx.training_result, ctx.trunk, ctx.world
    images_per_s = world * B * args.steps / dt
    ms_per_step = 1000.0 * dt / args.steps
    traffic = load_traffic()
    tsrc = traffic.get("source")

    default_cfg = (B == 1024 and Ng == 10000 and k == 100 and args.backbone == "resnet50" and args.backbone_dtype == "f32" and cl
                   and not args.no_fold_bn)

    def traffic_of(key, per_gpu_only=True):
        """HBM bytes of one steady-state step for kernel family `key` from the committed PMC profile of THIS workload (default
        arguments); None -- never a stale number -- for any other configuration."""
        e = traffic.get("kernels", {}).get(key)
        if not default_cfg or not isinstance(e, dict) or (world > 1 and not per_gpu_only):
            return None
        t = trunk.get(key)
        if t is not None and ksteps and e.get("launches") != t["n"] // ksteps:
            return None                    # the profile was taken with a different kernel dispatch: stale, not reported
        return e.get("bytes")

    line = {
        "metric": "images/sec descriptor extract + query x gallery search",
        "value": images_per_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: ResNet-50 fully-conv global descriptors + top-%d cosine search, "
                               "%d-row gallery shard per GPU, 224x224 synthetic images" % (k, Ng),
                   "images_per_gpu_per_step": B, "gallery_rows_per_gpu": Ng, "descriptor_dim": D, "k": k,
                   "backbone": args.backbone, "backbone_dtype": args.backbone_dtype, "activation_layout": "NHWC" if cl else "NCHW",
                   "bn_folded": not args.no_fold_bn, "parallelism": "gallery-row shards x%d + DP extraction" % world,
                   "collective_backend": backend, "ranks": world},
        "dist_per_s": images_per_s * Ng * world,
    }
    # whole step against the fp32 matrix-core peak (ResNet-50 convolutions + the distance GEMM; pooling / top-k are bytes, not FLOP)
    step_flop = RESNET50_GFLOP_PER_IMAGE * 1e9 * B + gemm_flop if args.backbone == "resnet50" else None
    if step_flop:
        line["roofline_step"] = {"bound": "mfma", "achieved": step_flop / (ms_per_step * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                                 "unit": "TFLOP/s", "frac": step_flop / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                 "algorithmic_flop_per_step_per_gpu": step_flop,
                                 "note": "%.2f GFLOP per image (ResNet-50 convolutions) x %d images + 2*M*N*D of the search" % (RESNET50_GFLOP_PER_IMAGE, B)}
    fam = {}
    kernel_names = {
        "isx_conv1x1_nhwc": "conv1x1_tail_kernel / cosine_gemm_kernel<ALIGNED, TM, TN, EPI = 2, BK> + conv1x1_stream_kernel for Cin = 64 (isx_conv1x1_nhwc: 1x1 convolutions as fp32-MFMA GEMMs over the pixels, bias/residual/ReLU fused)",
        "isx_conv1x1_dual_nhwc": "conv1x1_dual_tail_kernel / conv1x1_dual_nhwc_kernel (isx_conv1x1_dual_nhwc: last 1x1 conv + projection shortcut as one GEMM)",
        "isx_conv3x3_nhwc": "conv3x3_tail_kernel / conv3x3_nhwc_kernel (isx_conv3x3_nhwc: implicit GEMM, 128x128 tiles + 64x64 tail, bias/residual/ReLU fused)",
        "isx_conv3x3_expand_nhwc": "conv3x3_expand_kernel (isx_conv3x3_expand_nhwc: 3x3 convolution to 64 channels + 1x1 expansion + residual + ReLU, mid activation on chip)",
        "isx_stem7x7_pool_nhwc": "stem7x7_pool_kernel (isx_stem7x7_pool_nhwc: conv 7x7/2 + bias + ReLU + maxpool 3/2/1 as one kernel)",
    }
    for name, t in sorted(trunk.items()):
        mf = t["flop"] / (t["ms"] * 1e-3) / 1e12
        hb = t["bytes"] / (t["ms"] * 1e-3) / 1e9
        mfma_bound = t["flop"] / (PEAK_F32_MFMA_TFLOPS * 1e12) >= t["bytes"] / (PEAK_HBM_GBS * 1e9)
        o = {"kernel": kernel_names.get(name, name), "bound": "mfma" if mfma_bound else "hbm",
             "achieved": mf if mfma_bound else hb, "peak": PEAK_F32_MFMA_TFLOPS if mfma_bound else PEAK_HBM_GBS,
             "unit": "TFLOP/s" if mfma_bound else "GB/s",
             "frac": (mf / PEAK_F32_MFMA_TFLOPS) if mfma_bound else (hb / PEAK_HBM_GBS),
             "traffic": traffic_of(name), "traffic_unit": "HBM bytes per step (all launches of the family)",
             "traffic_over_algorithmic": (traffic_of(name) / (t["bytes"] / ksteps)) if traffic_of(name) else None,
             "traffic_source": tsrc if traffic_of(name) is not None else None,
             "launches_per_step": t["n"] // ksteps, "ms_per_step": t["ms"] / ksteps,
             "algorithmic_flop_per_step": t["flop"] / ksteps, "algorithmic_bytes_per_step": t["bytes"] / ksteps,
             "achieved_tflops": mf, "algorithmic_GBps": hb,
             # sum over launches of max(MFMA time, HBM time) / measured time: counts the HBM-bound layers of the family honestly
             "frac_of_per_launch_rooflines": t["floor_ms"] / t["ms"],
             # per layer shape (launches with the same algorithmic FLOP and bytes): which shapes sit furthest below their own roofline
             "shapes": sorted(({"launches_per_step": n_ // ksteps, "ms_per_launch": ms_sum / n_, "gflop": fl / 1e9, "mbytes": by / 1e6,
                                "tflops": fl / (ms_sum / n_ * 1e-3) / 1e12, "GBps": by / (ms_sum / n_ * 1e-3) / 1e9,
                                "frac_of_own_roofline": max(fl / (PEAK_F32_MFMA_TFLOPS * 1e9), by / (PEAK_HBM_GBS * 1e6)) / (ms_sum / n_),
                                "ms_above_roofline_per_step": (ms_sum / n_ - max(fl / (PEAK_F32_MFMA_TFLOPS * 1e9), by / (PEAK_HBM_GBS * 1e6))) * (n_ // ksteps)}
                               for (fl, by), (n_, ms_sum) in t["shapes"].items()), key=lambda e: -e["ms_above_roofline_per_step"]),
             "timing": "HIP events on the launch stream, %d instrumented steps after the timed region" % ksteps}
        fam[name] = o
    if gemm_ms is not None:
        tr = traffic_of("cosine_gemm", per_gpu_only=False)
        fam["cosine_gemm"] = {"kernel": "cosine_gemm_kernel<ALIGNED, TM, TN, EPI = 0, BK> (isx_cosine_sim)", "bound": "mfma",
                              "achieved": gemm_flop / (gemm_ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                              "frac": gemm_flop / (gemm_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                              "traffic": tr, "traffic_source": tsrc if tr is not None else None,
                              "launch_ms": gemm_ms, "algorithmic_flop_per_launch": gemm_flop, "shape": [M, Ng, D]}
        tr = traffic_of("gap_l2")
        fam["gap_l2"] = {"kernel": "gap_l2_nhwc_kernel (isx_gap_l2_nhwc)" if cl else "gap_l2_kernel (isx_gap_l2)", "bound": "hbm",
                         "achieved": gap_bytes / (gap_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": gap_bytes / (gap_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": tr,
                         "traffic_source": tsrc if tr is not None else None,
                         "launch_ms": gap_ms, "algorithmic_bytes_per_launch": gap_bytes}
    # `roofline` = the hand-written kernel family with the largest share of the step
    share = lambda o: o.get("ms_per_step", o.get("launch_ms", 0.0))
    if fam:
        dom = max(fam, key=lambda n: share(fam[n]))
        line["roofline"] = dict(fam[dom], share_of_step=share(fam[dom]) / ms_per_step, family=dom)
        # north_star's two NAMED kernels, inside the object the driver stores whole: the distance matmul with top-k ranking (the 10 k x 125 k x
        # 2048 shard of BASELINE configs[4] = one GPU's share of 10 k x 1 M at 8 GPUs, all-fp32 MFMA, end to end incl. the selection kernels; the
        # exact fp16-filter search of the same shard; the step's own 1024 x 10 k x 2048 GEMM) and the pooling kernel.  HIP events on the launch stream.
        hot = {}
        if isinstance(shard_result, dict) and "fp32_path" in shard_result:
            f32 = shard_result["fp32_path"]
            ms_e = f32.get("event_ms_this_rank") or f32["ms"]
            fl = 2.0 * shard_result["shape"][0] * shard_result["gallery_rows_per_gpu"] * shard_result["shape"][2]
            hot["cosine_topk_fp32"] = {"shape": [shard_result["shape"][0], shard_result["gallery_rows_per_gpu"], shard_result["shape"][2]], "k": k,
                                       "ms": ms_e, "tflops": fl / (ms_e * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                                       "frac": fl / (ms_e * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "bound": "mfma",
                                       "what": "isx_cosine_topk: fp32-MFMA score chunks + selection, end to end; 2*M*N*D FLOP credited"}
            ms_f = shard_result.get("event_ms_this_rank") or shard_result["ms"]
            hot["cosine_topk_fast"] = {"ms": ms_f, "identical": shard_result.get("identical_to_fp32_path"),
                                       "frac_of_f16_peak": fl / (ms_f * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS}
        if "cosine_gemm" in fam:
            hot["cosine_gemm_step"] = _pick(fam["cosine_gemm"], ("shape", "launch_ms", "achieved", "frac"))
        if "gap_l2" in fam:
            hot["gap_l2"] = {"GB_s": fam["gap_l2"]["achieved"], "frac": fam["gap_l2"]["frac"], "ms": fam["gap_l2"]["launch_ms"],
                             "bytes_per_image": gap_bytes / B, "traffic": fam["gap_l2"].get("traffic")}
        line["roofline"]["hot_kernels"] = hot
        line["roofline"]["traffic_profile"] = {"csrc_digest_of_profile": traffic.get("csrc_digest"), "csrc_digest_now": csrc_digest(),
                                               "fresh": bool(traffic.get("fresh"))}
        for n, o in fam.items():
            line["roofline_" + n] = o
    else:
        line["roofline"] = None
    if shard_result is not None:
        line["retrieval_shard"] = shard_result
    if regions_result is not None:
        line["extraction_regions"] = regions_result
    if ingest_result is not None:
        line["ingest_streaming"] = ingest_result
    if ingest_decode_result is not None:
        line["ingest_decode"] = ingest_decode_result
    if slab_result is not None:
        line["slab_roundtrip"] = slab_result
    if training_result is not None:
        line["training"] = training_result
    if world > 1:
        ex = exchange_legs or {}
        tot = sum(ex.values()) if ex else None
        line["exchange_ms"] = tot
        line["exchange"] = dict(ex, exposed_when_serialised_frac_of_step=(tot / ms_per_step if tot is not None else None),
                                overlapped=overlap, overlap_identical=overlap_identical,
                                implementation=(("isx_comm_allgather_rows | " if retrieval.exchange_backend(None, True).startswith("isx_") else
                                                 "torch.distributed all_gather_into_tensor | ") + retrieval.exchange_backend(None, True) + " + isx_topk_merge"),
                                communicators_in_data_path=1, merged_lists_identical_to_unsharded_search=merge_identical,
                                legs="query all-gather | per-shard top-k all-gather x 2 + isx_topk_merge"
                                     + (" (with the score GEMM and the top-k between them on a second stream, behind the next step's trunk)" if overlap else ""),
                                timing="HIP events on the launch stream, max over ranks, %d instrumented steps with the exchange in line" % ksteps)
    return line
