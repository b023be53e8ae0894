"""`cpu_baseline`: the reference's PyTorch-CPU path in modern torch on a bounded sample of the bench workload (kind "port"), timed on rank 0."""
import os
import sys
import time

from .common import ROOT, build_net, cpu_model, usable_cpus


def cpu_baseline(args, gallery_cpu, images_cpu):
    """The reference's PyTorch-CPU path in modern torch on a bounded sample of the same workload:
    fp32 eval-mode trunk -> mean over (H,W) -> x / sqrt(sum x^2 + 1e-10) -> torch.mm(q, G.t()) -> topk, every usable host
    core.  The oracle is NOT in the timed region; it only checks the sample's ranked lists afterwards."""
    import torch
    net = build_net(args.backbone, "f32", "cpu")
    threads = usable_cpus()
    torch.set_num_threads(threads)
    G = gallery_cpu

    def run(n):
        x = images_cpu[:n]
        with torch.no_grad():
            fmap = net.features(x)
            pooled = fmap.mean((2, 3))
            q = pooled / (pooled.pow(2).sum(1, keepdim=True) + 1e-10).sqrt()
            sim = torch.mm(q, G.t())
            return q, sim.topk(min(args.k, G.size(0)), dim=1)

    run(2)                                                  # warm caches / thread pool
    nb = images_cpu.size(0)
    passes, t0 = 0, time.time()
    while True:                                             # whole passes over the sample until ~cpu_seconds of work are done (1 .. 16 passes)
        q, (ts, ti) = run(nb)
        passes += 1
        dt = time.time() - t0
        if passes >= 16 or dt + dt / passes > args.cpu_seconds:
            break
    n = passes * nb
    checked = None
    try:                                                    # checker only, outside the timing
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import numpy as np
        import oracle as O
        _, oi = O.cosine_topk(q[:4].numpy(), G.numpy(), min(10, G.size(0)))
        checked = bool(np.array_equal(oi[:, 0], ti[:4, 0].numpy()))
    except Exception:
        pass
    return {"value": n / dt, "unit": "images/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "host_cpus_visible": os.cpu_count(),
            "sample": "%d images (%d passes over %d): torch-CPU fp32 %s trunk + mean-pool + L2 + torch.mm vs the %d-row gallery + topk(%d), "
                      "%d threads, %.1f s" % (n, passes, nb, args.backbone, G.size(0), args.k, threads, dt),
            "top1_matches_oracle_on_sample": checked}


def cpu_baseline_retrieval(args, seconds=6.0):
    """The retrieval half of the metric on the host cores (reference test/classif_finetune_test.py:82 + utils/metrics.py:25-55):
    `torch.mm(Q, G.t())` fp32 + `topk(k)` on a 1k x 62.5k x 2048 slice of BASELINE configs[4] (1/10 of the queries x 1/16 of the rows;
    distances/s is size-independent for a GEMM this large, so the figure is quoted per distance, not scaled), repeated for a bounded time,
    and the reference's literal per-rank Python AP loop (oracle.avg_precision_literal -- the checker's restatement, timed here as the
    CPU baseline only) on a few queries of a 10k-row gallery -> ms per query."""
    import torch
    threads = usable_cpus()
    torch.set_num_threads(threads)
    M, N, D, k = 1000, 62500, 2048, args.k
    g = torch.Generator().manual_seed(5)
    Q = torch.nn.functional.normalize(torch.randn(M, D, generator=g), dim=1)
    G = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    torch.mm(Q[:64], G.t()).topk(k, dim=1)                   # warm the thread pool
    t_mm = t_topk = 0.0
    reps, t0 = 0, time.time()
    while True:
        a = time.time()
        sim = torch.mm(Q, G.t())
        b = time.time()
        sim.topk(k, dim=1)
        c = time.time()
        t_mm += b - a; t_topk += c - b
        reps += 1
        el = time.time() - t0
        if reps >= 20 or el + el / reps > seconds:
            break
    out = {"value": reps * M * N / (t_mm + t_topk), "unit": "distances/s", "cores": threads, "kind": "port",
           "sample": "%d x (torch.mm + topk(%d)) on %d queries x %d rows x %d (a 1/10 x 1/16 slice of configs[4]), fp32, %d threads, %.1f s"
                     % (reps, k, M, N, D, threads, t_mm + t_topk),
           "mm_ms": 1e3 * t_mm / reps, "topk_ms": 1e3 * t_topk / reps, "mm_tflops": 2.0 * M * N * D * reps / t_mm / 1e12}
    try:                                                    # the literal rank-by-rank AP loop of the reference, on a 10k-row gallery
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        Ng, nq = 10000, 32
        lab_g = [i % (Ng // 10) for i in range(Ng)]
        simq = torch.mm(Q[:nq], G[:Ng].t())
        per_q, aps = [], []
        for i in range(nq):                                 # per-query times: the median is quoted (a single-threaded Python loop: scheduling noise moves the mean)
            a = time.perf_counter()
            aps.append(O.avg_precision_literal(simq[i], i % (Ng // 10), lab_g, 1, tensor_iteration=True))
            per_q.append(time.perf_counter() - a)
        per_q.sort()
        out["ap_loop_ms_per_query"] = 1e3 * per_q[nq // 2]
        out["ap_loop_ms_per_query_min_max"] = [1e3 * per_q[0], 1e3 * per_q[-1]]
        out["ap_loop_sample"] = "median of %d queries x %d gallery rows: sort + the per-rank Python loop of utils/metrics.py:25-45, walked over a torch index tensor as the reference does" % (nq, Ng)
        assert all(x is not None for x in aps)
    except Exception as e:
        out["ap_loop_error"] = "%s: %s" % (type(e).__name__, e)
    return out
