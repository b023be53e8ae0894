"""Side leg `retrieval_shard` of bench.py: BASELINE configs[4]'s per-GPU share -- 10 k replicated queries against a gallery sharded 125 k rows per GPU
(1 M rows at 8 GPUs): local top-k (exact fp16-filter search and the all-fp32 search: identical bits) + all-gather of the per-shard lists + merge,
end to end, and the full-rank average precision of the same queries without gathering the gallery (isx_ap_shard_*)."""
import json
import os
import sys
import time

from .common import (PEAK_F16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, RESNET50_GFLOP_PER_IMAGE, ROOT, build_net, load_traffic,
                     usable_cpus)


def measure(ctx):
    D, backend, dev, dist, ev, k, ops, rank, retrieval, torch, world = ctx.D, ctx.backend, ctx.dev, ctx.dist, ctx.ev, ctx.k, ctx.ops, ctx.rank, ctx.retrieval, ctx.torch, ctx.world
    Ms, Ns = 10000, 125000
    gq = torch.Generator(device=dev).manual_seed(1)
    Qs = ops.l2norm_rows(torch.randn(Ms, D, device=dev, generator=gq))              # same queries on every rank
    gg = torch.Generator(device=dev).manual_seed(100 + rank)
    Gs = ops.l2norm_rows(torch.randn(Ns, D, device=dev, generator=gg))

    event_ms = {}

    def time_search(fast):
        gal = retrieval.ShardedGallery(Gs, idx_base=rank * Ns, fast=fast)
        gal.search(Qs, k)                      # warm-up (fast: builds the cached fp16 image of the shard)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ts0 = time.perf_counter()
        ea_, eb_ = ev(), ev()
        ea_.record()
        for _ in range(3):
            res = gal.search(Qs, k)
        eb_.record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        ms_ = (time.perf_counter() - ts0) / 3 * 1e3
        if world > 1:
            tm_ = torch.tensor([ms_], device=dev, dtype=torch.float64)
            dist.all_reduce(tm_, op=dist.ReduceOp.MAX)
            ms_ = float(tm_.item())
        event_ms[fast] = ea_.elapsed_time(eb_) / 3      # HIP events on the launch stream around the three searches (this rank)
        return ms_, res

    ms32, (rs32, ri32) = time_search(False)      # every score on the fp32 matrix cores
    ms, (rs, ri) = time_search(True)             # fp16-MFMA filter + exact fp32 re-scoring: must be identical
    identical = bool(torch.equal(ri, ri32) and torch.equal(rs.view(torch.int32), rs32.view(torch.int32)))
    if not identical:                                # reported in the JSON line; never silently, never fatal for the headline number
        print("bench.py: WARNING isx_cosine_topk_fast differs from isx_cosine_topk on the shard workload", file=sys.stderr)
    assert ri.shape == (Ms, k) and int(ri.min()) >= 0 and int(ri.max()) < Ns * world
    flop = 2.0 * Ms * Ns * world * D
    # full-rank average precision of the same queries WITHOUT gathering the gallery (isx_ap_shard_*: the ranks of the positives are counts that
    # add over shards): labels as SURVEY 8d assigns them (row i of the whole gallery: i mod N / 10), 10 positives per query and shard
    L = Ns * world // 10
    glab_l = ((torch.arange(Ns, dtype=torch.int64) + rank * Ns) % L).to(torch.int32)
    qlab_l = (torch.arange(Ms, dtype=torch.int64) % L).to(torch.int32)
    gal32 = retrieval.ShardedGallery(Gs, idx_base=rank * Ns, fast=False)
    gal32.average_precisions(Qs, qlab_l, glab_l)             # warm-up at full size: the 5 GB score block comes out of the caching allocator afterwards
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ta0 = time.perf_counter()
    aps_ = gal32.average_precisions(Qs, qlab_l, glab_l)
    torch.cuda.synchronize()
    ap_ms = (time.perf_counter() - ta0) * 1e3
    if world > 1:
        tm_ = torch.tensor([ap_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tm_, op=dist.ReduceOp.MAX)
        ap_ms = float(tm_.item())
    ap_valid = aps_[aps_ == aps_]
    return {"shape": [Ms, Ns * world, D], "gallery_rows_per_gpu": Ns, "k": k, "ms": ms,
            "sharded_average_precision": {"ms": ap_ms, "queries": Ms, "mAP": float(ap_valid.mean()) if ap_valid.numel() else None,
                                          "includes": "fp32 score rows of the shard (isx_cosine_sim, query blocks) + isx_ap_shard_positives + _hist + isx_ap_from_hist"
                                                      + (" + all-gather of the positives' keys + all-reduce of the rank histograms" if world > 1 else "")},
            "dist_per_s": Ms * Ns * world / (ms * 1e-3),
            "tflops_end_to_end": flop / (ms * 1e-3) / 1e12,
            "frac_of_f16_mfma_peak": flop / (ms * 1e-3) / 1e12 / (PEAK_F16_MFMA_TFLOPS * world),
            "path": "isx_cosine_topk_fast (fp16-MFMA filter + exact fp32 re-scoring, bit-identical results)",
            "fp32_path": {"ms": ms32, "event_ms_this_rank": event_ms.get(False), "dist_per_s": Ms * Ns * world / (ms32 * 1e-3),
                          "tflops_end_to_end": flop / (ms32 * 1e-3) / 1e12,
                          "frac_of_f32_mfma_peak": flop / (ms32 * 1e-3) / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)},
            "identical_to_fp32_path": identical, "event_ms_this_rank": event_ms.get(True),
            "includes": "local top-k" + (" + %s all-gather of per-shard top-k + isx_topk_merge" % ("RCCL" if backend == "nccl" else backend) if world > 1 else "")}
