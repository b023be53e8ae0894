"""Side leg `slab_roundtrip` of bench.py."""
import json
import os
import sys
import time

from .common import (PEAK_F16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, RESNET50_GFLOP_PER_IMAGE, ROOT, build_net, load_traffic,
                     usable_cpus)


def measure(ctx):
    D, args, dev, k, ops, retrieval, torch = ctx.D, ctx.args, ctx.dev, ctx.k, ctx.ops, ctx.retrieval, ctx.torch
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
