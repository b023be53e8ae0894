#!/usr/bin/env python3
"""bench.py -- the hot path of BASELINE.json on N MI355X GPUs of one node.

This file: argument parsing, the self-launch, the set-up, THE STEP and the timed region (main()).  Everything else lives in bench_legs/:
common.py (peaks, the trunk under test, PMC traffic profile), regions / shard / ingest / slab / training (side legs, each `measure(ctx)`, outside the
timed region), report.py (rooflines per kernel family -> the full record), line.py (the driver's compact line), cpu_baseline.py.

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
    python bench.py --gpus 8                      # self-launching: starts its eight ranks itself (no launcher process)
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
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))

from bench_legs.common import (PEAK_F16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, RESNET50_GFLOP_PER_IMAGE, build_net, cpu_model, csrc_digest,  # noqa: E402,F401
                               load_traffic, usable_cpus)
from bench_legs.cpu_baseline import cpu_baseline, cpu_baseline_retrieval  # noqa: E402,F401
from bench_legs.line import MAX_LINE_BYTES, compact_line, write_detail  # noqa: E402,F401
from bench_legs import ingest as leg_ingest, regions as leg_regions, report, shard as leg_shard, slab as leg_slab, training as leg_training  # noqa: E402


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

    # everything a side leg needs from this run (bench_legs/*.measure(ctx)); the legs never touch the timed region
    ctx = types.SimpleNamespace(args=args, dev=dev, ev=ev, k=k, D=D, B=B, M=M, Ng=Ng, ops=ops, retrieval=retrieval, rank=rank, world=world, local=local, backend=backend,
                                dist=dist, torch=torch, net=net, cl=cl, synthetic_descriptors=synthetic_descriptors, synthetic_images=synthetic_images)

    def side_leg(leg, on_every_rank=False):
        """A side measurement never costs the headline: its failure is reported inside its own object (legs with collectives inside run unguarded on N > 1:
        every rank must take the same path)."""
        if on_every_rank and world > 1:
            return leg(ctx)
        try:
            return leg(ctx)
        except Exception as e:
            return {"error": "%s: %s" % (type(e).__name__, e)}

    if args.only_regions:                            # profiling aid: `rocprofv3 --kernel-trace -- python3 bench.py --only-regions` traces this leg alone
        out_r = leg_regions.measure(ctx)
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

    # ---- side legs (outside the timed region; bench_legs/) ---------------------------------------------------------------------------------
    ctx.dt = dt
    shard_result = None
    if not args.no_shard_bench:                      # BASELINE configs[4]'s per-GPU shard: 10 k queries x 125 k rows, exact-fast and all-fp32, sharded AP
        shard_result = side_leg(leg_shard.measure, on_every_rank=True)
        torch.cuda.empty_cache()
    ingest_result = ingest_decode_result = slab_result = training_result = None
    f32_folded = args.backbone_dtype == "f32" and not args.no_fold_bn
    if args.ingest_images > 0 and rank == 0 and f32_folded:          # a set that is not resident in HBM, PCIe-inclusive
        ingest_result = side_leg(leg_ingest.measure_streaming)
        torch.cuda.empty_cache()
    if args.slab_rows > 0 and rank == 0 and world == 1:              # gallery slab GPU -> file -> GPU (8f-2)
        slab_result = side_leg(leg_slab.measure)
    if args.decode_images > 0 and rank == 0 and f32_folded:          # extraction from JPEG files through the decoder farm (8f-4)
        ingest_decode_result = side_leg(leg_ingest.measure_decode)
        torch.cuda.empty_cache()
    if not args.no_train_bench and world == 1 and args.backbone_dtype == "f32":      # siamese training on one GPU (8f-1, configs[3])
        training_result = side_leg(leg_training.measure)
        torch.cuda.empty_cache()
    regions_result = None
    if not args.no_regions_bench:                    # BASELINE configs[2]: region-pooled descriptors at 448 x 448 + its 1 k x 100 k retrieval leg
        try:
            regions_result = leg_regions.measure(ctx)
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
        ctx.__dict__.update(exchange_legs=exchange_legs, gap_bytes=gap_bytes, gap_ms=gap_ms, gemm_flop=gemm_flop, gemm_ms=gemm_ms, ksteps=ksteps, trunk=trunk,
                            overlap=overlap, overlap_identical=overlap_identical, merge_identical=merge_identical, shard_result=shard_result,
                            ingest_result=ingest_result, ingest_decode_result=ingest_decode_result, slab_result=slab_result, training_result=training_result,
                            regions_result=regions_result)
        line = report.assemble(ctx)
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
