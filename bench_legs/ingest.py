"""Side legs `ingest_streaming` (a set that is NOT resident in HBM, PCIe-inclusive) and `ingest_decode` (extraction from JPEG files) of bench.py
(SURVEY 8f-4)."""
import json
import os
import sys
import time

from .common import (PEAK_F16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, RESNET50_GFLOP_PER_IMAGE, ROOT, build_net, load_traffic,
                     usable_cpus)


def measure_streaming(ctx):
    args, local, net, torch = ctx.args, ctx.local, ctx.net, ctx.torch
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


def measure_decode(ctx):
    B, args, dt, local, net, torch, world = ctx.B, ctx.args, ctx.dt, ctx.local, ctx.net, ctx.torch, ctx.world
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
