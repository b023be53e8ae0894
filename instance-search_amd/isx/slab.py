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


class SlabWriter(object):
    """Streaming writer: the header is written first (N and D are known: the extraction loop fills a slab of that shape), row blocks are appended
    as they come -- from the host or from the GPU, CHUNK_ROWS at a time through one reusable pinned buffer -- and `close` adds the labels and
    moves the file into place.  No copy of the whole array is ever made (round 4's save_slab held descriptors + `tobytes()`: 2 x 8.2 GB at 1 M x 2048)."""

    def __init__(self, path, rows, dim, has_labels=False, normalised=True):
        self.path, self.tmp, self.N, self.D = path, path + ".tmp", int(rows), int(dim)
        self.label_off = DATA_OFFSET + ((self.N * self.D * 4 + 4095) // 4096) * 4096 if has_labels else 0
        self.fd = os.open(self.tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        os.pwrite(self.fd, _HDR.pack(MAGIC, self.N, self.D, 1 if normalised else 0, self.label_off), 0)
        self.written = 0

    def _write(self, host, rows):
        """`rows` rows of the contiguous fp32 host tensor `host` behind the rows written so far (positional writes, READ_THREADS side by side)."""
        _pio(self.fd, host, DATA_OFFSET + self.written * self.D * 4, rows * self.D * 4, write=True)
        self.written += rows

    def append(self, block):
        """block: (n, D) fp32 rows that follow the rows written so far (torch tensor on any device, or ndarray)."""
        if not isinstance(block, torch.Tensor):
            a = np.ascontiguousarray(block, dtype=np.float32)
            if a.ndim != 2 or a.shape[1] != self.D:
                raise ValueError("slab rows must be (n, %d), got %s" % (self.D, a.shape))
            block = torch.from_numpy(a)
        block = block.detach()
        if block.dim() != 2 or block.size(1) != self.D or block.dtype != torch.float32:
            raise ValueError("slab rows must be (n, %d) float32, got %s %s" % (self.D, tuple(block.shape), block.dtype))
        if self.written + block.size(0) > self.N:
            raise ValueError("more rows appended (%d) than the slab was opened for (%d)" % (self.written + block.size(0), self.N))
        if not block.is_cuda:
            if block.size(0):
                self._write(block.contiguous(), block.size(0))
            return
        # from the GPU: pieces of READ_CHUNK_BYTES through the two pinned staging buffers -- the device-to-host copy of piece c + 1 runs
        # while piece c goes to the file
        rows_per = max(1, READ_CHUNK_BYTES // (self.D * 4))
        stage = _read_stage(rows_per * self.D)
        with torch.cuda.device(block.device):
            starts = list(range(0, block.size(0), rows_per))
            done = [None, None]

            def fetch(c):
                r0 = starts[c]
                piece = block[r0:r0 + rows_per]
                stage[c & 1][: piece.numel()].view(piece.size(0), self.D).copy_(piece, non_blocking=True)
                done[c & 1] = torch.cuda.Event()
                done[c & 1].record()
                return piece.size(0)

            sizes = {}
            if starts:
                sizes[0] = fetch(0)
            for c in range(len(starts)):
                done[c & 1].synchronize()
                if c + 1 < len(starts):
                    sizes[c + 1] = fetch(c + 1)
                self._write(stage[c & 1], sizes[c])

    def close(self, labels=None):
        if self.written != self.N:
            os.close(self.fd)
            os.unlink(self.tmp)
            raise ValueError("slab opened for %d rows, %d appended" % (self.N, self.written))
        try:
            if self.label_off:
                lab = labels.detach().cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
                lab = np.ascontiguousarray(lab, dtype=np.int32)
                if lab.shape != (self.N,):
                    raise ValueError("labels must have shape (%d,)" % self.N)
                os.pwrite(self.fd, lab.tobytes(), self.label_off)
            elif labels is not None:
                raise ValueError("the slab was opened without labels")
            else:
                os.ftruncate(self.fd, DATA_OFFSET + self.N * self.D * 4)        # an empty slab still has its header page
        finally:
            os.close(self.fd)
        os.replace(self.tmp, self.path)


def save_slab(path, descriptors, labels=None, normalised=True):
    """Write (N, D) fp32 descriptors (torch tensor on any device, or ndarray) and optional int labels, row block by row block (SlabWriter)."""
    N, D = descriptors.shape
    w = SlabWriter(path, N, D, has_labels=labels is not None, normalised=normalised)
    w.append(descriptors if isinstance(descriptors, torch.Tensor) else np.asarray(descriptors))
    w.close(labels)


def save_gallery(path, descriptors, ref_set, labels):
    """A gallery as the evaluation entry points hold it -- the descriptor slab of `ref_set` ((image, label, path) tuples) and the run's sorted
    label list -- as a slab file + `<path>.labels.json` (label names: the slab stores their indices)."""
    import json
    ids = dict((lab, i) for i, lab in enumerate(labels))
    save_slab(path, descriptors, torch.tensor([ids[lab] for _, lab, _ in ref_set], dtype=torch.int32))
    with open(path + ".labels.json", "w") as f:
        json.dump({"labels": list(labels), "paths": [p for _, _, p in ref_set]}, f)


def load_gallery(path, device="cpu", rows=None):
    """(descriptors, ref_set, labels) of save_gallery: ref_set = [(None, label, path)] -- what the metrics read of a gallery item."""
    import json
    desc, lab = load_slab(path, device, rows)
    with open(path + ".labels.json") as f:
        side = json.load(f)
    names, paths = side["labels"], side["paths"]
    lo = 0 if rows is None else rows[0]
    ref_set = [(None, names[int(i)], paths[lo + j]) for j, i in enumerate(lab.tolist())]
    return desc, ref_set, names


READ_CHUNK_BYTES = 64 << 20      # pinned staging pieces of a GPU load (two of them, kept for the process)
READ_THREADS = 4
_STAGE = []
_READ_POOL = None


def _read_stage(floats):
    """Two pinned staging buffers of at least `floats` floats (allocated once: pinning 2 x 64 MB costs as much as reading 1 GB)."""
    if not _STAGE or _STAGE[0].numel() < floats:
        del _STAGE[:]
        _STAGE.extend(torch.empty((floats,), dtype=torch.float32).pin_memory() for _ in range(2))
    return _STAGE


def _pio(fd, buf, offset, nbytes, write=False):
    """nbytes between the file at `offset` and the contiguous host tensor `buf`: READ_THREADS positional reads (or writes) side by side (the
    page cache <-> user copy of one thread runs at 4-7 GB/s; mmap adds a fault per 4 KB page on top)."""
    global _READ_POOL
    if nbytes <= 0:
        return
    view = memoryview(buf.numpy()).cast("B")
    if _READ_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _READ_POOL = ThreadPoolExecutor(max_workers=READ_THREADS)
    piece = -(-nbytes // READ_THREADS)
    piece = (piece + 4095) // 4096 * 4096

    def go(a):
        b, at = min(a + piece, nbytes), a
        while at < b:
            k = os.pwritev(fd, [view[at:b]], offset + at) if write else os.preadv(fd, [view[at:b]], offset + at)
            if k <= 0:
                raise IOError("descriptor slab: short %s at byte %d" % ("write" if write else "read (the file is shorter than its header says)", offset + at))
            at += k

    list(_READ_POOL.map(go, range(0, nbytes, piece)))


def _pread_into(fd, buf, offset, nbytes):
    _pio(fd, buf, offset, nbytes)


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
            rows_per = max(1, min(max(n, 1), READ_CHUNK_BYTES // (D * 4)))
            stage = _read_stage(rows_per * D)
            events = [None, None]
            copy_stream = torch.cuda.current_stream(dev)
            fd = os.open(path, os.O_RDONLY)
            try:
                for c, r0 in enumerate(range(0, n, rows_per)):
                    r1 = min(r0 + rows_per, n)
                    b = c & 1
                    if events[b] is not None:
                        events[b].synchronize()             # the previous copy out of this buffer is done
                    _pread_into(fd, stage[b], DATA_OFFSET + (lo + r0) * D * 4, (r1 - r0) * D * 4)
                    out[r0:r1].copy_(stage[b][: (r1 - r0) * D].view(r1 - r0, D), non_blocking=True)
                    events[b] = torch.cuda.Event()
                    events[b].record(copy_stream)
            finally:
                os.close(fd)
            torch.cuda.synchronize(dev)
    labels = None
    if info["has_labels"]:
        lm = np.memmap(path, dtype=np.int32, mode="r", offset=info["label_offset"] + lo * 4, shape=(n,)) if n else np.zeros((0,), np.int32)
        labels = torch.from_numpy(np.array(lm)).to(dev)
    return out, labels
