#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on N MI355X GPUs of one node.

Workload (BASELINE configs[1], "ResNet-50 fully-conv global descriptors, 10k-image synthetic
gallery"): one STEP = one batch of synthetic 224x224 images through the whole path
    images --ResNet-50 trunk (fp32; the fused 7x7 stem, every 1x1 / 3x3 convolution of the residual blocks and every epilogue: libisx; MIOpen runs nothing)-->
           (B,2048,7,7) feature map --isx_gap_l2 (HIP)--> L2-normalised 2048-d descriptors
           --[N>1: RCCL all-gather of the query descriptors]--
           --isx_cosine_sim (fp32 MFMA) + isx_topk_rows (HIP) against this rank's gallery shard-->
           --[N>1: RCCL all-gather of per-shard top-k + isx_topk_merge]--> ranked top-100 lists
Inputs (images, gallery slab) are resident in HBM before the timed region.  Weak scaling: every
rank extracts B images and holds a 10k-row gallery shard.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8                      # self-launching: starts torch.distributed.run itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 3

Rank 0 prints the record twice: first `{"bench_detail": {...}}` -- everything (per-family and per-layer-shape rooflines, every
side measurement in full), also written to `bench_detail.json` (gpurun_out/ when that directory exists, else next to this
file) -- and then, as the LAST stdout line, the compact record the driver parses (<= 6 KB: metric / value / ms_per_step / config /
ONE `roofline` for the dominant kernel family / `roofline_step` / `retrieval_shard` / `extraction_regions` / `cpu_baseline`).
The K timed steps carry no per-kernel instrumentation; the per-kernel HIP-event timings behind the `roofline*` objects come
from a separate instrumented pass over the same step.

N > 1: the search stage of step i (all-gather of the query descriptors, score GEMM against this rank's shard, top-k, all-gather of the
per-shard lists + isx_topk_merge) is issued on a second HIP stream behind an event and rides behind the trunk of step i + 1
(`--no-overlap-exchange` keeps everything on one stream); `exchange_ms` is what the query all-gather, the two result all-gathers
and the merge cost when nothing hides them (HIP events in the instrumented pass), `overlap_identical` says that both
schedules returned the same bits.  `cpu_baseline` is timed on
rank 0 at every N (the other ranks wait at the barrier).

Side objects of the same line (each bounded to a few seconds; a failure in one is reported inside it and never costs the
headline): `retrieval_shard` = BASELINE configs[4]'s per-GPU shard (10k x 125k x 2048) end to end; `extraction_regions` =
BASELINE configs[2]: ResNet-50 TuneClassifSub on 448 x 448 images -> best-location class-score descriptors (reference
train/classif_regions.py:107-132) with its own roofline, plus the 1k x 100k retrieval + metrics leg of that config.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))

PEAK_F16_MFMA_TFLOPS = 2500.0    # dense fp16/bf16 MFMA peak of one MI355X (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0            # HBM3E spec
RESNET50_GFLOP_PER_IMAGE = 8.17  # 2 x 4.087 GMAC, convolutions of the 224x224 trunk (~8.2)


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
                    help="layout of the backbone activations (same fp32 math); isx_gap_l2 consumes either layout in place")
    ap.add_argument("--no-fold-bn", action="store_true",
                    help="keep BatchNorm as separate kernels (default: folded into the convolutions for inference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shard-bench", action="store_true", help="skip the 10k x 125k retrieval-shard side measurement")
    ap.add_argument("--no-kernel-pass", action="store_true", help="skip the instrumented per-kernel pass (roofline objects of the trunk)")
    ap.add_argument("--no-regions-bench", action="store_true", help="skip the BASELINE configs[2] side measurement (region path at 448 x 448)")
    ap.add_argument("--only-regions", action="store_true", help="run the configs[2] side measurement alone and print it (profiling aid; single GPU)")
    ap.add_argument("--regions-batch", type=int, default=128, help="448 x 448 images per launch of the region path")
    ap.add_argument("--no-overlap-exchange", action="store_true",
                    help="N > 1: keep the search stage of a step (query all-gather, GEMM, top-k, result exchange) on the main stream "
                         "(default: on a second stream behind the next step's trunk)")
    ap.add_argument("--slab-rows", type=int, default=131072, help="rows of the descriptor-slab round trip (GPU -> file -> GPU, isx/slab.py); 0 skips it")
    ap.add_argument("--decode-images", type=int, default=16384, help="images of the decode-inclusive ingest leg (JPEG files -> descriptors); 0 skips it")
    ap.add_argument("--ingest-images", type=int, default=65536,
                    help="images of the non-resident streaming-ingest side measurement (0 = skip); host memory: 4096 distinct uint8 images, the rest views")
    ap.add_argument("--no-train-bench", action="store_true", help="skip the siamese-training side measurement (BASELINE configs[3] on one GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` from a bare command line: start the N ranks as CHILD processes of a parent that has
    not touched the GPU (no torch import yet), one `python bench.py ...` per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its
    environment -- what `torch.distributed.run` would set, without a launcher process in between (the launcher opens the GPU too, and a
    GPU box of this pool admits six processes per card) -- relay their output and exit with the first non-zero code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: required for RCCL across processes on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    env.update(WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(args.gpus)]
    code = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0 and code == 0:
                code = rc
                for q in pending:                      # a rank failed: the others would wait in a collective for ever
                    q.terminate()
        if pending:
            time.sleep(0.05)
    return code


def build_net(name, dtype, device, channels_last=False, fold_bn=False):
    import torch
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


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


# ---- the driver's line ------------------------------------------------------------------------------------------------
MAX_LINE_BYTES = 6144


def _r(x, nd=4):
    """numbers to `nd` significant digits (the side file keeps full precision)"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    if isinstance(x, dict):
        return {k_: _r(v, nd) for k_, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def _pick(d, keys):
    return {k_: d[k_] for k_ in keys if isinstance(d, dict) and k_ in d}


def compact_line(full, detail_file=None):
    """The full record -> the line the driver parses: the contract's scalar fields verbatim, ONE `roofline` (dominant kernel family),
    a one-number-per-family table, and the headline numbers of the side measurements.  Everything else stays in the side file.
    Guaranteed <= MAX_LINE_BYTES: optional objects are dropped (largest first) if a future field ever pushes it over."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                        "dtype", "data", "config", "dist_per_s"))
    ro = full.get("roofline")
    if isinstance(ro, dict):
        line["roofline"] = _pick(ro, ("family", "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_unit",
                                      "traffic_over_algorithmic", "traffic_source", "share_of_step", "launches_per_step", "ms_per_step",
                                      "algorithmic_flop_per_step", "algorithmic_bytes_per_step", "frac_of_per_launch_rooflines", "timing", "hot_kernels",
                                      "traffic_profile"))
        if isinstance(line["roofline"].get("traffic_source"), str):
            line["roofline"]["traffic_source"] = line["roofline"]["traffic_source"].split(":")[0]
    else:
        line["roofline"] = None
    if isinstance(full.get("roofline_step"), dict):
        line["roofline_step"] = _pick(full["roofline_step"], ("bound", "achieved", "peak", "unit", "frac", "algorithmic_flop_per_step_per_gpu"))
    fams = {}
    for key, o in full.items():
        if key.startswith("roofline_") and key != "roofline_step" and isinstance(o, dict):
            fams[key[len("roofline_"):]] = [o.get("bound"), o.get("frac"), o.get("ms_per_step", o.get("launch_ms"))]
    if fams:
        line["families"] = {"columns": ["bound", "frac", "ms_per_step"], "rows": fams}
    sh = full.get("retrieval_shard")
    if isinstance(sh, dict):
        c = _pick(sh, ("error", "shape", "gallery_rows_per_gpu", "k", "ms", "dist_per_s", "tflops_end_to_end", "frac_of_f16_mfma_peak",
                       "identical_to_fp32_path", "includes"))
        if isinstance(sh.get("fp32_path"), dict):
            c["fp32_path"] = _pick(sh["fp32_path"], ("ms", "dist_per_s", "frac_of_f32_mfma_peak"))
        if isinstance(sh.get("sharded_average_precision"), dict):
            c["sharded_average_precision"] = _pick(sh["sharded_average_precision"], ("ms", "queries", "mAP"))
        line["retrieval_shard"] = c
    rg = full.get("extraction_regions")
    if isinstance(rg, dict):
        c = _pick(rg, ("error", "images_per_s", "ms_per_launch", "images_per_launch", "all_convolutions_in_libisx", "images_per_s_all_gpus", "note"))
        if isinstance(rg.get("roofline"), dict):
            c["roofline"] = _pick(rg["roofline"], ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic"))
        if isinstance(rg.get("retrieval_1000x100000"), dict):
            c["retrieval_1000x100000"] = _pick(rg["retrieval_1000x100000"], ("descriptor_dim", "total_ms", "dist_per_s",
                                                                              "cosine_sim_frac_of_f32_mfma_peak", "mAP"))
        c["workload"] = "BASELINE configs[2]: ResNet-50 TuneClassifSub @448x448 -> best-location descriptors"
        line["extraction_regions"] = c
    ig = full.get("ingest_streaming")
    if isinstance(ig, dict):
        line["ingest_streaming"] = _pick(ig, ("error", "images", "resident_images_per_s", "extract_pcie_inclusive_images_per_s", "streamed_over_resident",
                                              "descriptors_identical"))
    dg = full.get("ingest_decode")
    if isinstance(dg, dict):
        line["ingest_decode"] = _pick(dg, ("error", "images", "cores", "images_per_s", "decode_only_images_per_s", "decode_bound", "descriptors_identical_to_decode_first"))
    sl = full.get("slab_roundtrip")
    if isinstance(sl, dict):
        line["slab_roundtrip"] = _pick(sl, ("error", "rows", "write_GB_per_s", "read_GB_per_s", "identical", "search_identical"))
    tr = full.get("training")
    if isinstance(tr, dict):
        line["training"] = _pick(tr, ("error", "reference_config_triplets_per_s", "frozen_trunk_triplets_per_s", "reference_over_frozen", "reference_config", "statistic",
                                      "reference_config_with_prefix_cache_triplets_per_s"))
        if isinstance(tr.get("roofline"), dict):
            line["training"]["roofline"] = _pick(tr["roofline"], ("bound", "achieved", "peak", "unit", "frac", "ms_per_step", "algorithmic_flop_per_step"))
    if "exchange_ms" in full:
        line["exchange_ms"] = full["exchange_ms"]
        line["exchange"] = _pick(full.get("exchange") or {}, ("query_allgather_ms", "result_allgather_merge_ms", "exposed_when_serialised_frac_of_step",
                                                              "overlapped", "overlap_identical", "implementation", "communicators_in_data_path",
                                                              "merged_lists_identical_to_unsharded_search"))
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("error", "value", "unit", "cores", "kind", "sample", "cpu_model", "top1_matches_oracle_on_sample"))
        if isinstance(cb.get("retrieval"), dict):
            c["retrieval"] = _pick(cb["retrieval"], ("error", "value", "unit", "cores", "kind", "sample", "mm_tflops", "ap_loop_ms_per_query",
                                                      "ap_loop_sample", "ap_loop_error"))
        line["cpu_baseline"] = c
    if detail_file:
        line["detail_file"] = detail_file
    line = _r(line)
    for key in ("value", "ms_per_step", "dist_per_s"):          # the contract's scalars keep their digits
        if key in full:
            line[key] = full[key]
    for victim in ("families", "slab_roundtrip", "training", "ingest_decode", "ingest_streaming", "exchange", "extraction_regions", "retrieval_shard", "roofline_step"):      # never expected: a safety net
        if len(json.dumps(line)) <= MAX_LINE_BYTES:
            break
        line.pop(victim, None)
        line.setdefault("dropped_for_size", []).append(victim)
    return line


def write_detail(full):
    """The full record -> bench_detail.json (gpurun_out/ when present: that directory travels back from the GPU box)."""
    out_dir = os.path.join(ROOT, "gpurun_out")
    path = os.path.join(out_dir if os.path.isdir(out_dir) else ROOT, "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT)
    except Exception:
        return None


def csrc_digest():
    """sha256 (16 hex digits) over the kernel sources (instance-search_amd/csrc/*.{hip,hpp,cpp}, Makefile, include/isx.h): what a PMC profile is a
    profile OF.  profiles/summarize_prof.py stamps it into roofline_traffic.json; a bench run whose sources hash differently reports the
    traffic as null (`traffic_stale`) instead of bytes that belong to other kernels.  (The GPU box has no .git: a content hash, not a commit.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "instance-search_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(csrc, "*.cpp"))
                       + [os.path.join(csrc, "Makefile"), os.path.join(ROOT, "include", "isx.h")]):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def load_traffic():
    """HBM bytes measured with rocprofv3 PMC passes on an EARLIER run of this command (profiles/roofline_traffic.json, written
    by profiles/summarize_prof.py): a property of that profiled run, stamped with its source -- never of the run printing it."""
    path = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        t = json.load(open(path))
    except Exception:
        return {}
    try:
        t["fresh"] = bool(t.get("csrc_digest")) and t.get("csrc_digest") == csrc_digest()
    except Exception:
        t["fresh"] = False
    if not t["fresh"]:                               # the kernels changed since the counters were read: no bytes rather than stale bytes
        t = {"source": t.get("source"), "csrc_digest": t.get("csrc_digest"), "fresh": False, "kernels": {}, "regions_leg": {}}
    return t


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    if torch.get_num_threads() > usable_cpus():                  # a box that shows 256 CPUs and owns 16: an OpenMP team of 128 is throttled into the ground
        torch.set_num_threads(usable_cpus())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the HIP path)")
    # debugging aid for boxes with ONE GPU: ISX_BENCH_ONE_DEVICE=1 maps every rank to cuda:0 and uses gloo, so the N > 1 code path
    # (query all-gather, per-shard search, result all-gather, merge) can be exercised; the numbers it prints mean nothing
    one_device = world > 1 and os.environ.get("ISX_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs cuda:%d but only %d device(s) are visible" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        backend = "gloo" if one_device else "nccl"
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if os.environ.get("ISX_BENCH_MIOPEN_FIND", "0") == "1":      # A/B: let MIOpen benchmark its algorithms for the layers it still runs
        torch.backends.cudnn.benchmark = True
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

    # side measurement: BASELINE configs[2] -- ResNet-50 region-pooled descriptors (classif_regions path) on 448 x 448 images,
    # then that config's retrieval leg at 1k queries x 100k gallery rows of those (class-score, 464-d) descriptors
    def regions_bench():
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

    if args.only_regions:                            # profiling aid: `rocprofv3 --kernel-trace -- python3 bench.py --only-regions` traces this leg alone
        out_r = regions_bench()
        if rank == 0:
            print(json.dumps({"extraction_regions": out_r}), flush=True)
        return


    overlap = world > 1 and not args.no_overlap_exchange
    side = torch.cuda.Stream(device=dev) if world > 1 else None
    if world > 1 and backend == "nccl":
        # ONE communicator carries both all-gathers of the step (query rows, then the per-shard lists): libisx's own, opened here -- a collective, so
        # every rank reaches it together, before any step.  If it cannot be opened on every rank the ranks agree (all-reduce MIN inside
        # native_comm_for, one line on stderr) and BOTH all-gathers go through torch.distributed's communicator instead: still one communicator
        # in the data path, and a scaling number rather than none.  ISX_REQUIRE_NATIVE_COMM=1 turns that into a non-zero exit.
        if retrieval.exchange_backend(None, True).startswith("isx_") and retrieval.native_comm_for(None) is None \
                and os.environ.get("ISX_REQUIRE_NATIVE_COMM", "0") == "1":
            raise SystemExit("bench.py: libisx's RCCL communicator could not be opened on every rank (ISX_REQUIRE_NATIVE_COMM=1)")
    exch_ev = []

    def exchange(s, i):
        """per-shard lists of every rank -> merged global lists (RCCL all-gather x 2 + isx_topk_merge)"""
        all_s, all_i = retrieval.exchange_topk(s, i)          # RCCL group: isx_shard_topk_allgather (one grouped launch on this stream)
        return ops.topk_merge(all_s, all_i)

    # deferred mode (N > 1): everything behind the descriptors of a step -- query all-gather, score GEMM, top-k, result all-gathers, merge -- is
    # issued on the side stream behind an event, with its own score buffer and two alternating descriptor buffers; the main stream goes
    # straight on to the trunk of the next step, which hides the collectives' latencies and the search's low-occupancy tails
    q_bufs = [torch.empty((B, D), device=dev) for _ in range(2)] if world > 1 else None
    q_free = [None, None]                          # side-stream event: the query all-gather that read buffer b has completed
    sim_side = torch.empty((M, Ng), device=dev) if world > 1 else None
    slot_box = [0]

    def search(Q_local, sim_buf, timed, marks):
        """query block of every rank -> scores against this rank's shard -> per-shard top-k [-> exchange + merge]"""
        if timed:
            x0, x1 = ev(), ev(); x0.record()
        Q = retrieval.gather_queries(Q_local)
        if timed:
            x1.record()
            c, d = ev(), ev(); c.record()
        ops.cosine_sim(Q, shard, out=sim_buf)
        if timed:
            d.record(); gemm_ev.append((c, d))
        s, i = ops.topk_rows(sim_buf, k, idx_base=gallery.idx_base)
        if world > 1:
            if timed:
                x2, x3 = ev(), ev(); x2.record()
            s, i = exchange(s, i)
            if timed:
                x3.record(); exch_ev.append((x0, x1, x2, x3))
        return s, i

    def step(timed, deferred=False):
        """One step.  `timed`: the instrumented pass (HIP events around the pool, the GEMM and the exchange legs on torch's current
        stream, which is the stream every libisx launch goes to).  `deferred` (N > 1): the search stage of the step runs on the side stream."""
        with torch.no_grad():
            if args.backbone_dtype == "bf16":
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    fmap = net.features(images)
                fmap = fmap.float()
            else:
                fmap = net.features(images)
        if world > 1 and deferred:
            b2 = slot_box[0]
            slot_box[0] ^= 1
            if q_free[b2] is not None:
                torch.cuda.current_stream().wait_event(q_free[b2])      # the search two steps back has read this descriptor buffer
            ops.gap_l2(fmap, out=q_bufs[b2])
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(side):
                side.wait_event(ready)
                out = search(q_bufs[b2], sim_side, False, None)
                q_free[b2] = torch.cuda.Event()
                q_free[b2].record(side)
            return out
        if timed:
            a, b = ev(), ev(); a.record()
        ops.gap_l2(fmap, out=q_local)
        if timed:
            b.record(); gap_ev.append((a, b))
        return search(q_local, sim, timed, None)

    for _ in range(args.warmup):
        step(False, overlap)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(False, overlap)
    torch.cuda.synchronize()                     # every stream of the device: the deferred exchange of the last step is inside the timed region
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert out[1].shape == (M, k) and int(out[1].min()) >= 0
    overlap_identical = None
    merge_identical = None
    if world > 1:
        # both schedules on the same inputs: the deferred exchange must return the bits of the in-line one
        o1 = step(False, False)
        o2 = step(False, True)
        torch.cuda.synchronize()
        same = bool(torch.equal(o1[1], o2[1]) and torch.equal(o1[0].view(torch.int32), o2[0].view(torch.int32)))
        flag = torch.tensor([1 if same else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        overlap_identical = bool(int(flag.item()) == 1)
        # the merged lists of the P shards against ONE unsharded search over the whole gallery (every rank's shard gathered; galleries up to 200 k rows:
        # the rehearsal sizes and the default 10 k rows per GPU at any N <= 8): the canonical order makes them the same bits
        if Ng * world <= 200000:
            whole = torch.empty((world * Ng, D), device=dev)
            if backend == "nccl":
                dist.all_gather_into_tensor(whole, shard)
            else:
                parts = [torch.empty((Ng, D)) for _ in range(world)]
                dist.all_gather(parts, shard.cpu())
                whole = torch.cat(parts, 0).to(dev)
            q_all = retrieval.gather_queries(q_local)
            us, ui = ops.cosine_topk(q_all, whole, k)
            same = bool(torch.equal(ui, o1[1]) and torch.equal(us.view(torch.int32), o1[0].view(torch.int32)))
            flag = torch.tensor([1 if same else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            merge_identical = bool(int(flag.item()) == 1)
            del whole

    # ---- instrumented pass (outside the timed region): per-launch HIP events of the hand-written kernels ----------
    trunk, ksteps = {}, 0
    gemm_ms = gap_ms = None
    exchange_legs = None
    if not args.no_kernel_pass or world > 1:          # N > 1: the exchange legs are always measured
        ksteps = max(1, min(args.steps, 5))
        ops.KERNEL_TIMER = [] if not args.no_kernel_pass else None
        for _ in range(ksteps):
            step(True)
        torch.cuda.synchronize()
        timer, ops.KERNEL_TIMER = ops.KERNEL_TIMER or [], None
        gemm_ms = sum(a.elapsed_time(b) for a, b in gemm_ev) / len(gemm_ev)
        gap_ms = sum(a.elapsed_time(b) for a, b in gap_ev) / len(gap_ev)
        if exch_ev:
            q_ms = sum(e[0].elapsed_time(e[1]) for e in exch_ev) / len(exch_ev)
            r_ms = sum(e[2].elapsed_time(e[3]) for e in exch_ev) / len(exch_ev)
            em = torch.tensor([q_ms, r_ms], device=dev, dtype=torch.float64)
            dist.all_reduce(em, op=dist.ReduceOp.MAX)
            exchange_legs = {"query_allgather_ms": float(em[0]), "result_allgather_merge_ms": float(em[1])}
        for name, flop, nbytes, ea, eb in timer:
            t = trunk.setdefault(name, {"n": 0, "flop": 0.0, "bytes": 0.0, "ms": 0.0, "floor_ms": 0.0, "shapes": {}})
            ms_ = ea.elapsed_time(eb)
            t["n"] += 1; t["flop"] += flop; t["bytes"] += nbytes; t["ms"] += ms_
            sh = t["shapes"].setdefault((flop, nbytes), [0, 0.0])            # launches of one layer shape share (FLOP, bytes)
            sh[0] += 1; sh[1] += ms_
            # per-launch roofline floor: the slower of the MFMA time and the HBM time of that launch's algorithmic work
            t["floor_ms"] += max(flop / (PEAK_F32_MFMA_TFLOPS * 1e9), nbytes / (PEAK_HBM_GBS * 1e6))
        if world > 1:
            dist.barrier()
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

    if not args.no_shard_bench:
        if world > 1:
            shard_result = shard_bench()             # collective inside: every rank must take the same path
        else:
            try:
                shard_result = shard_bench()
            except Exception as e:                   # a failed side measurement must not cost the headline number
                shard_result = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    # side measurement: PCIe-inclusive extraction of a set that is NOT resident in HBM (train/_common.BatchStager: two pinned
    # buffers + a copy stream, batch i + 1 stacked and copied while batch i runs) against the same set resident -- rank 0 only, no collective
    ingest_result = None

    def ingest_bench():
        from train import _common as TC
        from train import classif_finetune as cf
        n, blk = args.ingest_images, 4096
        gi = torch.Generator().manual_seed(7)
        block = torch.randint(0, 256, (min(blk, n), 224, 224, 3), dtype=torch.uint8, generator=gi)      # decoded RGB images as the raw ingest carries them
        data = [(block[i % block.size(0)], "l%d" % (i % 100), "p%d" % i) for i in range(n)]              # n per-image host tensors (the reference's dataset form)
        P = cf.P
        saved, budget = dict(P.__dict__), TC.RESIDENT_BUDGET_BYTES
        TC.RAW_INGEST["mean"], TC.RAW_INGEST["std"] = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
        try:
            P.cuda_device, P.embeddings_classify, P.embeddings_fc7, P.test_pre_proc, P.test_batch_size = local, False, False, True, 64

            def timed_pass():
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                slab = cf.get_embeddings(net, data, local, 2048)
                torch.cuda.synchronize()
                return time.perf_counter() - t0_, slab
            TC.drop_resident()
            t_up, _ = timed_pass()                      # uploads the set (one-time) + extracts
            t_res, slab_res = timed_pass()              # resident: batches are device-side row gathers
            TC.drop_resident()
            TC.RESIDENT_BUDGET_BYTES = 0                # nothing may stay in HBM: every batch crosses PCIe
            t_str, slab_str = timed_pass()
            same = bool(torch.equal(slab_res, slab_str))
        finally:
            TC.drop_resident()
            TC.RESIDENT_BUDGET_BYTES = budget
            TC.RAW_INGEST["mean"] = TC.RAW_INGEST["std"] = None
            P.__dict__.clear(); P.__dict__.update(saved)
        return {"images": n, "image_bytes": 224 * 224 * 3, "ingest": "uint8 (H,W,3) host tensors, normalised on the device (isx_images_u8_to_f32)",
                "resident_images_per_s": n / t_res, "extract_pcie_inclusive_images_per_s": n / t_str, "streamed_over_resident": t_res / t_str,
                "first_pass_with_upload_images_per_s": n / t_up, "descriptors_identical": same,
                "path": "train.classif_finetune.get_embeddings -> train._common.BatchStager (2 pinned buffers, copy stream, look-ahead 1)"}

    def ingest_decode_bench():
        """Extraction FROM FILES: 2048 JPEG files (224 x 224, smooth pattern + noise, quality 90) written to a scratch folder, a 16 384-entry gallery
        cycling through them as train._common.LazyImage entries, through get_embeddings (decode pool -> pinned staging -> copy stream -> trunk).
        The same files decoded by the pool alone give the host's decode rate: whichever is lower bounds an evaluation run on a real folder."""
        import shutil
        import tempfile
        from concurrent.futures import ThreadPoolExecutor
        import numpy as np
        from PIL import Image
        from test import _common as C
        from train import _common as TC
        from train import classif_finetune as cf
        n_files, n = 2048, args.decode_images
        tmp = tempfile.mkdtemp(prefix="isx_decode_")
        rng = np.random.default_rng(3)

        def write(i):
            low = rng.integers(0, 256, (8, 8, 3), dtype=np.uint8) if False else np.random.default_rng(i).integers(0, 256, (8, 8, 3), dtype=np.uint8)
            im = np.asarray(Image.fromarray(low).resize((224, 224), Image.BICUBIC), dtype=np.int16)
            im = np.clip(im + np.random.default_rng(10 ** 6 + i).integers(-12, 13, im.shape), 0, 255).astype(np.uint8)
            Image.fromarray(im).save(os.path.join(tmp, "%05d.jpg" % i), quality=90)

        workers = TC.decode_workers()
        try:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                list(pool.map(write, range(n_files)))
            file_bytes = sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp)) / float(n_files)
            load = C.ImageLoader(raw=True)
            files = [os.path.join(tmp, "%05d.jpg" % (i % n_files)) for i in range(n)]
            from train import _decode_farm as DF
            farm = DF.decode_farm()
            if farm is not None:                                   # decoder processes alone: files -> shared slots, nothing copied out
                for t in farm.submit(files[:256], 224 * 224 * 3):
                    t.tensor(); t.release()
                t0_ = time.perf_counter()
                pending = [farm.submit(files[a:a + 512], 224 * 224 * 3) for a in range(0, 4096, 512)]
                for tickets in pending:
                    for t in tickets:
                        t.tensor(); t.release()
                decode_only = 4096 / (time.perf_counter() - t0_)
                workers = farm.n
            else:
                t0_ = time.perf_counter()
                with ThreadPoolExecutor(max_workers=workers) as pool:
                    for _ in pool.map(load, files[:4096]):
                        pass
                decode_only = 4096 / (time.perf_counter() - t0_)
            first = load(files[0])
            data = [(TC.LazyImage(f, load, first.shape, first.dtype), "l%d" % (i % 100), f) for i, f in enumerate(files)]
            P = cf.P
            saved = dict(P.__dict__)
            TC.RAW_INGEST["mean"], TC.RAW_INGEST["std"] = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
            try:
                P.cuda_device, P.embeddings_classify, P.embeddings_fc7, P.test_pre_proc, P.test_batch_size = local, False, False, True, 64
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                slab = cf.get_embeddings(net, data, local, 2048)
                torch.cuda.synchronize()
                t_pipe = time.perf_counter() - t0_
                # the same files decoded up front (the reference's way), then extracted from RAM: descriptors must be identical
                eager = [(load(f), lab, f) for _, lab, f in data[:1024]]
                TC.drop_resident()
                slab_e = cf.get_embeddings(net, eager, local, 2048)
                same = bool(torch.equal(slab[:1024], slab_e))
            finally:
                TC.drop_resident()
                TC.RAW_INGEST["mean"] = TC.RAW_INGEST["std"] = None
                P.__dict__.clear(); P.__dict__.update(saved)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        rate = n / t_pipe
        ips = world * B * args.steps / dt                      # the headline rate of this run (inputs resident in HBM)
        return {"images": n, "files": n_files, "format": "JPEG 224x224 quality 90, %.0f KB per file, PIL decode" % (file_bytes / 1e3), "cores": workers,
                "decoders": "processes (train/_decode_farm.py)" if farm is not None else "threads",
                "images_per_s": rate, "decode_only_images_per_s": decode_only, "decode_bound": bool(rate < 0.9 * ips),
                "fraction_of_resident_rate": rate / ips, "descriptors_identical_to_decode_first": same,
                "path": "test._common.load_sets(lazy) form: LazyImage -> decoder processes (3 batches ahead, shared slots) -> BatchStager pinned staging -> copy stream -> trunk"}

    if args.ingest_images > 0 and rank == 0 and args.backbone_dtype == "f32" and not args.no_fold_bn:
        try:
            ingest_result = ingest_bench()
        except Exception as e:
            ingest_result = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    def slab_bench():
        """Next-scope row f2: a gallery slab GPU -> file (SlabWriter: row blocks through one pinned buffer) -> GPU (mmap -> pinned staging -> HBM),
        and a search against the re-read gallery.  The rates are the box's file system's as much as the code's; the bits must be the same."""
        import tempfile
        from isx import slab as _slab
        n = args.slab_rows
        g_ = torch.Generator(device=dev).manual_seed(11)
        desc = ops.l2norm_rows(torch.randn((n, D), device=dev, generator=g_))
        lab = (torch.arange(n, dtype=torch.int32) % 1000)
        tmp = tempfile.mkdtemp(prefix="isx_slab_")
        path = os.path.join(tmp, "gallery.slab")
        try:
            torch.cuda.synchronize(); t0_ = time.perf_counter()
            _slab.save_slab(path, desc, lab)
            t_w = time.perf_counter() - t0_
            t0_ = time.perf_counter()
            back = retrieval.ShardedGallery.from_slab(path, dev)
            torch.cuda.synchronize()
            t_r = time.perf_counter() - t0_
            same = bool(torch.equal(back.shard, desc))
            q = desc[:256].clone()
            s1, i1 = retrieval.ShardedGallery(desc, idx_base=0).search(q, k)
            s2, i2 = back.search(q, k)
            same_search = bool(torch.equal(i1, i2) and torch.equal(s1, s2))
            nbytes = os.path.getsize(path)
        finally:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
        return {"rows": n, "dim": D, "file_bytes": nbytes, "write_GB_per_s": nbytes / t_w / 1e9, "read_GB_per_s": nbytes / t_r / 1e9, "identical": same,
                "search_identical": same_search, "where": tempfile.gettempdir(),
                "path": "isx.slab.save_slab (SlabWriter, streamed from HBM) -> isx.retrieval.ShardedGallery.from_slab (mmap -> pinned -> HBM)"}

    slab_result = None
    if args.slab_rows > 0 and rank == 0 and world == 1:
        try:
            slab_result = slab_bench()
        except Exception as e:
            slab_result = {"error": "%s: %s" % (type(e).__name__, e)}

    ingest_decode_result = None
    if args.decode_images > 0 and rank == 0 and args.backbone_dtype == "f32" and not args.no_fold_bn:
        try:
            ingest_decode_result = ingest_decode_bench()
        except Exception as e:
            ingest_decode_result = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    # side measurement: next-scope row f1 (BASELINE configs[3] on ONE GPU) -- siamese triplet training of DescriptorNet(ResNet-50) on the
    # reference's configuration (layer4 + head trained) and with the whole trunk frozen: tools/bench_train.py, 2 epochs each (N = 1 only)
    training_result = None
    if not args.no_train_bench and world == 1 and args.backbone_dtype == "f32":
        try:
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
            training_result = {"workload": "BASELINE configs[3] on one GPU: DescriptorNet(ResNet-50, 2048) triplet training with per-epoch hard-negative mining, "
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
        except Exception as e:
            training_result = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    regions_result = None
    if not args.no_regions_bench:
        try:
            regions_result = regions_bench()
        except Exception as e:                       # never costs the headline; every rank catches alike (the collectives inside are bracketed by it)
            ops.KERNEL_TIMER = None
            regions_result = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
        if world > 1:                                # every rank gets here: slowest rank's launch time, and whether all ranks succeeded
            ok = "error" not in regions_result
            tm_ = torch.tensor([regions_result["ms_per_launch"] if ok else 0.0, 0.0 if ok else 1.0], device=dev, dtype=torch.float64)
            dist.all_reduce(tm_, op=dist.ReduceOp.MAX)
            if ok and float(tm_[1]) == 0.0:
                regions_result["ms_per_launch_max_over_ranks"] = float(tm_[0])
                regions_result["images_per_s_all_gpus"] = world * regions_result["images_per_launch"] / (float(tm_[0]) * 1e-3)
            elif ok:
                regions_result["note"] = "another rank failed this side measurement: rank 0's own numbers only"

    if rank == 0:
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
        if not args.no_cpu_baseline:                 # rank 0 at every N; the other ranks wait at the barrier below
            try:
                line["cpu_baseline"] = cpu_baseline(args, shard.cpu(), images_cpu)
            except Exception as e:
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                line["cpu_baseline"]["retrieval"] = cpu_baseline_retrieval(args)
            except Exception as e:
                line["cpu_baseline"]["retrieval"] = {"error": "%s: %s" % (type(e).__name__, e)}
        detail_file = write_detail(line)
        print(json.dumps({"bench_detail": line}), flush=True)          # everything, on an EARLIER line (and in the side file)
        print(json.dumps(compact_line(line, detail_file)), flush=True)      # the driver's line: last on stdout, <= 6 KB
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        retrieval.close_native_comms()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
