"""On-disk format of a descriptor slab.  The reference never persists descriptors
(it re-extracts the gallery on every run, test/classif_finetune_test.py:80-81); a 1 M x 2048 gallery
is 8.2 GB of extraction work worth keeping.

Layout (little endian):
    0    8   magic  b"ISXSLAB1"
    8    8   N      rows (u64)
    16   4   D      columns (u32)
    20   4   flags  bit 0: rows are L2-normalised
    24   8   label_offset (u64, 0 = no labels)
    32  32   reserved
    4096     N*D float32, row-major          (page aligned: mmap -> pinned staging -> HBM)
    label_offset   N int32 labels

`load_slab(..., rows=(lo, hi))` maps only a row range, which is how the ranks of a sharded gallery
(isx.retrieval.shard_bounds) each read their own contiguous 1/P of the file.
"""
import os
import struct

import numpy as np
import torch

MAGIC = b"ISXSLAB1"
DATA_OFFSET = 4096
_HDR = struct.Struct("<8sQIIQ32x")
CHUNK_ROWS = 1 << 16            # 512 MB staging pieces at D = 2048


def save_slab(path, descriptors, labels=None, normalised=True):
    """Write (N, D) fp32 descriptors (torch tensor on any device, or ndarray) and optional int labels."""
    d = descriptors.detach().cpu().numpy() if isinstance(descriptors, torch.Tensor) else np.asarray(descriptors)
    d = np.ascontiguousarray(d, dtype=np.float32)
    N, D = d.shape
    label_off = 0
    if labels is not None:
        lab = labels.detach().cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
        lab = np.ascontiguousarray(lab, dtype=np.int32)
        if lab.shape != (N,):
            raise ValueError("labels must have shape (%d,)" % N)
        label_off = DATA_OFFSET + ((N * D * 4 + 4095) // 4096) * 4096
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(_HDR.pack(MAGIC, N, D, 1 if normalised else 0, label_off))
        f.seek(DATA_OFFSET)
        f.write(d.tobytes())
        if labels is not None:
            f.seek(label_off)
            f.write(lab.tobytes())
    os.replace(tmp, path)


def slab_info(path):
    with open(path, "rb") as f:
        magic, N, D, flags, label_off = _HDR.unpack(f.read(_HDR.size))
    if magic != MAGIC:
        raise ValueError("%s is not a descriptor slab (bad magic)" % path)
    return {"rows": N, "dim": D, "normalised": bool(flags & 1), "has_labels": label_off != 0, "label_offset": label_off}


def load_slab(path, device="cpu", rows=None):
    """(descriptors (n, D) fp32 tensor on `device`, labels (n,) int32 tensor or None) for the row range
    `rows` = (lo, hi) (default: all).  GPU loads stream through a pinned staging buffer."""
    info = slab_info(path)
    N, D = info["rows"], info["dim"]
    lo, hi = (0, N) if rows is None else rows
    if not (0 <= lo <= hi <= N):
        raise ValueError("row range (%d, %d) outside [0, %d]" % (lo, hi, N))
    n = hi - lo
    mm = np.memmap(path, dtype=np.float32, mode="r", offset=DATA_OFFSET + lo * D * 4, shape=(n, D)) if n else np.zeros((0, D), np.float32)
    dev = torch.device(device)
    if dev.type == "cpu":
        out = torch.from_numpy(np.array(mm))
    else:
        with torch.cuda.device(dev):                        # copies, events and the final sync all on the TARGET device's stream
            out = torch.empty((n, D), dtype=torch.float32, device=dev)
            stage = [torch.empty((min(CHUNK_ROWS, max(n, 1)), D), dtype=torch.float32).pin_memory() for _ in range(2)]
            events = [None, None]
            copy_stream = torch.cuda.current_stream(dev)
            for c, r0 in enumerate(range(0, n, CHUNK_ROWS)):
                r1 = min(r0 + CHUNK_ROWS, n)
                b = c & 1
                if events[b] is not None:
                    events[b].synchronize()                 # the previous copy out of this buffer is done
                stage[b][: r1 - r0].copy_(torch.from_numpy(np.ascontiguousarray(mm[r0:r1])))
                out[r0:r1].copy_(stage[b][: r1 - r0], non_blocking=True)
                events[b] = torch.cuda.Event()
                events[b].record(copy_stream)
            torch.cuda.synchronize(dev)
    labels = None
    if info["has_labels"]:
        lm = np.memmap(path, dtype=np.int32, mode="r", offset=info["label_offset"] + lo * 4, shape=(n,)) if n else np.zeros((0,), np.int32)
        labels = torch.from_numpy(np.array(lm)).to(dev)
    return out, labels
