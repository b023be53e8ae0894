"""Shared pieces of the four per-approach modules: batch staging and net construction."""
import os

import numpy as np
import torch

from isx import backbones


# Raw ingest: a dataset may carry the DECODED images as (H,W,3) uint8 tensors instead of normalised
# fp32 tensors; stage_batch then ships the uint8 batch (a quarter of the bytes over PCIe) and applies
# ToTensor + Normalize on the GPU (libisx isx_images_u8_to_f32).  test/_common.load_sets fills this in.
RAW_INGEST = {"mean": None, "std": None}


def normalise_u8_batch(x_u8, device):
    """(B,H,W,3) uint8 -> (B,3,H,W) fp32, (x/255 - mean) / std; HIP kernel on the GPU, same arithmetic in torch on the CPU."""
    mean, std = RAW_INGEST["mean"], RAW_INGEST["std"]
    if mean is None:
        raise RuntimeError("uint8 images in the dataset but no mean/std registered (train._common.RAW_INGEST)")
    if device >= 0:
        from isx import ops
        src = getattr(x_u8, "_isx_rows", None)
        x_u8 = x_u8 if x_u8.is_cuda else x_u8.pin_memory().cuda(non_blocking=True)
        y = ops.images_u8_to_f32(x_u8.contiguous(), mean, std, channels_last=True)
        if src is not None:
            y._isx_rows = src                   # which rows of which resident set these images are (model/siamese.py prefix-feature cache)
        return y
    x = x_u8.permute(0, 3, 1, 2).float().div_(255.0)
    m = torch.tensor(list(mean)).view(1, -1, 1, 1)
    s_ = torch.tensor(list(std)).view(1, -1, 1, 1)
    return (x - m) / s_


# Lazy ingest: the reference decodes a whole folder into a Python list before the first launch
# (test/classif_finetune_test.py:62-73, utils/image.py:211-214); a 1 M-image gallery is 150 GB of decoded pixels -- more than the host has -- and
# the GPU waits for the last file before it sees the first.  A folder dataset can instead carry LazyImage entries: the usual (image, label, path)
# tuples whose image is decoded when a batch needs it, by a farm of decoder processes (train/_decode_farm.py), a bounded number of batches ahead
# of the trunk (BatchStager) and dropped once stacked into the pinned staging buffer.  Labels and paths are plain as ever.
_DECODE_POOL = None


def decode_workers():
    from utils.general import usable_cpus
    return int(os.environ.get("ISX_DECODE_THREADS", "0")) or min(16, usable_cpus())


def _decode_pool():
    global _DECODE_POOL
    if _DECODE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _DECODE_POOL = ThreadPoolExecutor(max_workers=max(1, decode_workers()))
    return _DECODE_POOL


class LazyImage(object):
    """Stands where the image tensor of a (image, label, path) tuple stands; `shape` / `dtype` are those of the decoded image (all images of a
    lazy set share them -- the reference's datasets are pre-sized, train/global_p.py image_sizes), `get()` returns the tensor.
    A loader that reads plain 8-bit RGB (`load.farm_kind == "rgb_u8"`, test/_common.ImageLoader) is served by the decoder PROCESSES of
    train/_decode_farm.py -- get() is then a view of a shared slot, valid until release(); any other loader runs on the thread pool."""
    __slots__ = ("path", "load", "shape", "dtype", "_fut")
    is_cuda = False

    def __init__(self, path, load, shape, dtype):
        self.path, self.load, self.shape, self.dtype, self._fut = path, load, tuple(shape), dtype, None

    def _check(self, t):
        if tuple(t.shape) != self.shape or t.dtype != self.dtype:
            raise RuntimeError("lazy ingest needs same-sized images: %s is %s %s, the set's first image %s %s (ISX_LAZY_INGEST=0 decodes ragged folders up front)"
                               % (self.path, tuple(t.shape), t.dtype, self.shape, self.dtype))
        return t

    def _decode(self):
        return self._check(self.load(self.path))

    def farmed(self):
        return getattr(self.load, "farm_kind", None) == "rgb_u8" and getattr(self.load, "raw", False) and self.dtype == torch.uint8

    def prefetch(self):
        if self._fut is None:
            prefetch_all([self])

    def shared(self):
        """True when get() returns a view of a decoder slot (to be copied out before release())."""
        return self._fut is not None and not hasattr(self._fut, "cancel")

    def get(self):
        self.prefetch()
        return self._check(self._fut.result()) if self.shared() else self._fut.result()

    def release(self):
        if self._fut is not None and self.shared():
            self._fut.release()
        self._fut = None


def prefetch_all(ims):
    """Hand every not-yet-requested LazyImage of `ims` to the decoders: one submission to the process farm for the plain RGB loaders (chunks of
    files per worker), the thread pool for the rest."""
    todo = [im for im in ims if isinstance(im, LazyImage) and im._fut is None]
    farmed = [im for im in todo if im.farmed()]
    if farmed:
        from ._decode_farm import decode_farm
        farm = decode_farm()
        if farm is not None:
            by_bytes = {}
            for im in farmed:
                by_bytes.setdefault(im.shape[0] * im.shape[1] * im.shape[2], []).append(im)
            for nbytes, group in by_bytes.items():
                for im, t in zip(group, farm.submit([im.path for im in group], nbytes)):
                    im._fut = t
    for im in todo:
        if im._fut is None:
            im._fut = _decode_pool().submit(im._decode)


def is_lazy(dataset):
    return len(dataset) > 0 and isinstance(dataset[0][0], LazyImage)


def resolve_images(ims):
    """Tensors of a list of images, LazyImage entries decoded (in parallel) and released."""
    if not any(isinstance(im, LazyImage) for im in ims):
        return ims
    prefetch_all(ims)
    out = []
    for im in ims:
        if isinstance(im, LazyImage):
            t = im.get()
            out.append(torch.from_numpy(t.numpy().copy()) if im.shared() else t)   # a decoder slot is recycled after release(); numpy's memcpy, not torch's OpenMP team
        else:
            out.append(im)
    for im in ims:
        if isinstance(im, LazyImage):
            im.release()
    return out


# Resident datasets: the reference keeps a dataset as a Python list of per-image CPU tensors and stacks + copies a batch
# at every step (train/siamese_descriptor.py:95-137, train/classif_finetune.py:92-104); on a 288 GB GPU the whole set
# (uint8 pixels with the raw ingest, or fp32 tensors) is copied to HBM ONCE and batches are row gathers on the device.
# Profile of a siamese training epoch before: 70 % of the wall time in CPU torch.stack + pageable H2D copies.
RESIDENT_BUDGET_BYTES = 48 << 30
_RESIDENT = []


_STACK_POOL = None


def _parallel_stack(chunk, out):
    """out[:len(chunk)] = stack(chunk): one memcpy per image (numpy's copy loop: the GIL is released, no OpenMP team), spread over a few threads.
    torch.stack moved the same bytes at 1.75 GB/s next to busy decoder processes (44 ms per 128 images of 448 x 448 x 3): its OpenMP team of
    16 has to assemble on cores the decoders occupy."""
    global _STACK_POOL
    k = len(chunk)
    first = chunk[0]
    if first.is_cuda or out.is_cuda or not first.is_contiguous() or any(im.shape != first.shape or im.dtype != first.dtype for im in chunk):
        torch.stack(chunk, 0, out=out[:k])
        return
    dst = out[:k].numpy()
    srcs = [im.numpy() for im in chunk]
    workers = min(4, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    if workers <= 1 or k < 8 or k * first.numel() * first.element_size() < (4 << 20):
        for j, a in enumerate(srcs):
            np.copyto(dst[j], a)
        return
    if _STACK_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _STACK_POOL = ThreadPoolExecutor(max_workers=workers)
    step = -(-k // workers)

    def part(j0):
        for j in range(j0, min(j0 + step, k)):
            np.copyto(dst[j], srcs[j])

    for f in [_STACK_POOL.submit(part, j0) for j0 in range(0, k, step)]:
        f.result()


class ResidentImages(object):
    def __init__(self, images, device):
        self.images = list(images)                  # strong references: the ids below stay valid
        self.index = dict((id(im), i) for i, im in enumerate(self.images))
        first = self.images[0]
        dev = torch.device('cuda', device)
        self.data = torch.empty((len(self.images),) + tuple(first.shape), dtype=first.dtype, device=dev)
        # chunks of 256 images are stacked straight into one of two reusable PINNED buffers (a fresh pageable 150 MB tensor per chunk cost
        # 0.19 s in page faults: 5.8 of the 9.4 s of a 4400-image test run) and copied asynchronously; an event per buffer guards its reuse
        n = min(256, len(self.images))
        with torch.cuda.device(dev):
            stage = [torch.empty((n,) + tuple(first.shape), dtype=first.dtype).pin_memory() for _ in range(2 if len(self.images) > n else 1)]
            done = [None, None]
            for c, i in enumerate(range(0, len(self.images), n)):
                b = c & 1 if len(stage) == 2 else 0
                if done[b] is not None:
                    done[b].synchronize()
                chunk = self.images[i:i + n]
                _parallel_stack(chunk, stage[b])
                self.data[i:i + len(chunk)].copy_(stage[b][:len(chunk)], non_blocking=True)
                done[b] = torch.cuda.Event()
                done[b].record(torch.cuda.current_stream(dev))
            torch.cuda.current_stream(dev).synchronize()

    def covers(self, ims):
        return all(id(im) in self.index for im in ims)

    def gather(self, ims):
        idx = torch.tensor([self.index[id(im)] for im in ims], dtype=torch.int64, device=self.data.device)
        x = self.data.index_select(0, idx)
        x._isx_rows = (self, idx)               # provenance: lets a frozen trunk prefix look its features up instead of recomputing them
        return x

    def normalised_rows(self, rows):
        """The network input of the given rows of the set (fp32, normalised)."""
        x = self.data.index_select(0, rows)
        return normalise_u8_batch(x, self.data.device.index) if x.dtype == torch.uint8 else x


def make_resident(dataset, device):
    """Register the images of `dataset` ((tensor, label, path) tuples) as a device-resident block; no-op on the CPU, for
    ragged image sizes, or beyond RESIDENT_BUDGET_BYTES."""
    if device < 0 or not dataset or is_lazy(dataset):
        return None
    ims = [im for im, _, _ in dataset]
    if any(r.covers(ims) for r in _RESIDENT):
        return None
    first = ims[0]
    if any(im.shape != first.shape or im.dtype != first.dtype or im.is_cuda for im in ims):
        return None
    if len(ims) * first.numel() * first.element_size() > RESIDENT_BUDGET_BYTES:
        return None
    r = ResidentImages(ims, device)
    _RESIDENT.append(r)
    del _RESIDENT[:-4]                              # keep the four most recent sets (train / gallery / queries / one spare)
    return r


def drop_resident():
    del _RESIDENT[:]


def stage_images(ims, device):
    """A list of same-shaped image tensors (fp32 CHW, or uint8 HWC for the raw ingest) -> one device batch, normalised."""
    ims = resolve_images(ims)
    if device >= 0:
        for r in _RESIDENT:
            if r.covers(ims):
                x = r.gather(ims)
                return normalise_u8_batch(x, device) if x.dtype == torch.uint8 else x
    x = torch.stack(ims, 0)
    if x.dtype == torch.uint8:
        return normalise_u8_batch(x, device)
    if device >= 0 and not x.is_cuda:
        x = x.pin_memory().cuda(non_blocking=True)
    return x


class BatchStager(object):
    """The consecutive batches of a (tensor, label, path) dataset on the device, in dataset order -- what fold_batches hands to a
    get_embeddings `run` function.  Three regimes:

      resident      the set lives in HBM (make_resident): a batch is a device-side row gather                     (stage_images)
      streaming     GPU, same-shaped untransformed images, set NOT resident (beyond RESIDENT_BUDGET_BYTES: a 1 M-image gallery is
                    150 GB as uint8): two reusable PINNED host buffers + two device buffers + a copy stream.  `get(i)` returns batch i
                    (already in flight) and, before returning, stacks batch i + 1 into the other pinned buffer and enqueues its H2D copy
                    on the copy stream -- host stacking and PCIe of batch i + 1 overlap the trunk of batch i (and of i - 1, still running
                    when get(i) is called).  The reference stacks + copies every batch on the compute path
                    (train/classif_finetune.py:92-104).  uint8 images are normalised on the device after the copy (isx_images_u8_to_f32).
      plain         CPU runs, transformed or ragged images: stage_batch as before.

    Same values whatever the regime (tests/test_gpu_dropin.py: streamed descriptors == resident descriptors, bit for bit)."""

    def __init__(self, dataset, batch_size, trans, device, force_streaming=None):
        self.dataset, self.bs, self.trans, self.device = dataset, int(batch_size), trans, device
        self.streaming = False
        self.lazy = is_lazy(dataset)
        if device >= 0 and trans is None and self.bs > 0 and len(dataset) > 0 and torch.cuda.is_available():
            ims = [im for im, _, _ in dataset]
            first = ims[0]
            uniform = not any(tuple(im.shape) != tuple(first.shape) or im.dtype != first.dtype or im.is_cuda for im in ims)
            resident = not self.lazy and any(r.covers(ims) for r in _RESIDENT)
            self.streaming = uniform and not resident if force_streaming is None else bool(force_streaming) and uniform
        if self.streaming:
            dev = torch.device('cuda', device)
            n = min(self.bs, len(dataset))
            shape, dtype = (n,) + tuple(first.shape), first.dtype
            self.dev = dev
            self.copy_stream = torch.cuda.Stream(device=dev)
            self.host = [torch.empty(shape, dtype=dtype).pin_memory() for _ in range(2)]
            self.devbuf = [torch.empty(shape, dtype=dtype, device=dev) for _ in range(2)]
            self.copied = [None, None]          # copy-stream event: the H2D copy out of host[b] into devbuf[b] has completed
            self.consumed = [None, None]        # compute-stream event: the kernels reading devbuf[b] have been enqueued (and, once reached, run)
            self.inflight = {}                  # batch start index -> (slot, rows)

    def _submit(self, start):
        slot = (start // self.bs) & 1
        chunk = [im for im, _, _ in self.dataset[start:start + self.bs]]
        if self.lazy:
            # decode-ahead: the files of this batch and of the next two are with the decoder threads before this call blocks on the first image --
            # batch i + 2 is being decoded while batch i + 1 is stacked / copied and the trunk runs batch i; at most three batches are decoded in RAM
            prefetch_all([im for im, _, _ in self.dataset[start:start + 3 * self.bs]])
            lazy_chunk, chunk = chunk, [im.get() for im in chunk]
        if self.copied[slot] is not None:
            self.copied[slot].synchronize()                      # the pinned buffer is free once its previous copy has left it
        _parallel_stack(chunk, self.host[slot])
        if self.lazy:
            for im in lazy_chunk:
                im.release()                                     # decoder slots go back once their pixels are in the pinned buffer
        with torch.cuda.stream(self.copy_stream):
            if self.consumed[slot] is not None:
                self.copy_stream.wait_event(self.consumed[slot])  # the device buffer is free once its previous readers have run
            self.devbuf[slot][:len(chunk)].copy_(self.host[slot][:len(chunk)], non_blocking=True)
            self.copied[slot] = torch.cuda.Event()
            self.copied[slot].record(self.copy_stream)
        self.inflight[start] = (slot, len(chunk))

    def get(self, start, batch):
        """Device batch of dataset[start:start + len(batch)] (normalised fp32; channels-last for the uint8 ingest)."""
        if not self.streaming:
            return stage_batch(batch, self.trans, self.device)
        cur = torch.cuda.current_stream(self.dev)
        prev = start - self.bs
        if prev >= 0:                                             # the consumers of the previous batch are enqueued by now: its device buffer may be refilled behind them
            pslot = (prev // self.bs) & 1
            self.consumed[pslot] = torch.cuda.Event()
            self.consumed[pslot].record(cur)
        if start not in self.inflight:
            self._submit(start)
        nxt = start + self.bs
        if nxt < len(self.dataset) and nxt not in self.inflight:
            self._submit(nxt)                                     # look-ahead: stacked + copied while batch `start` (and start - bs) compute
        slot, rows = self.inflight.pop(start)
        assert rows == len(batch)
        cur.wait_event(self.copied[slot])
        x = self.devbuf[slot][:rows]
        if x.dtype == torch.uint8:
            return normalise_u8_batch(x, self.device)             # reads the device buffer into a fresh fp32 tensor
        return x


def stage_batch(batch, trans, device):
    """Stack the (already normalised unless `trans` is given) images of a batch and move them."""
    if trans is None:
        return stage_images([im for im, _, _ in batch], device)
    return stage_images([trans(im) for im, _, _ in batch], device)


# Images per trunk launch on the GPU.  The reference's test batch size (64, train/global_p.py) sizes a 12 GB card; in eval mode the
# descriptor of an image does not depend on the batch it rides in, so on an MI355X (288 GB) the get_embeddings functions raise the batch
# to a multiple of the caller's size holding at least this many pixels -- 512 images of 224 x 224 -- which is where the convolution
# launches fill the chip (bench.py: 14.8 k images/s at 256 images per launch, 15.6 k at 512).  0 = the caller's size as given.
_MIN_DEVICE_BATCH_PIXELS = int(os.environ.get("ISX_MIN_DEVICE_BATCH_PIXELS", str(512 * 224 * 224)))


def _image_pixels(shape):
    """H * W of an image tensor: (C,H,W) fp32 tensors (C = 1 or 3 leading) or raw (H,W,3) uint8 images (3 trailing)."""
    if len(shape) != 3:
        return None
    if shape[2] == 3 and shape[0] not in (1, 3):
        return shape[0] * shape[1]
    return shape[1] * shape[2] if shape[0] in (1, 3) else shape[0] * shape[1]


def device_batch_size(P, dataset, shape=None):
    """Images per trunk launch for images of `shape` (default: the first image's): the caller's P.test_batch_size times the smallest
    integer that brings the launch to _MIN_DEVICE_BATCH_PIXELS.  Ragged datasets ask per shape bucket (fold_shape_buckets)."""
    bs = int(P.test_batch_size)
    if bs <= 0 or P.cuda_device < 0 or _MIN_DEVICE_BATCH_PIXELS <= 0 or len(dataset) == 0 or not torch.cuda.is_available():
        return bs
    pixels = _image_pixels(tuple(dataset[0][0].shape) if shape is None else tuple(shape))
    if pixels is None:
        return bs
    return bs * max(1, -(-_MIN_DEVICE_BATCH_PIXELS // max(bs * pixels, 1)))


def fold_shape_buckets(f, dataset, batch_size, stage=None):
    """Call f(indices, items) on batches of SAME-SHAPED images: the dataset is bucketed by image shape (first-seen order,
    dataset order inside a bucket), every bucket cut into batches of `batch_size` -- an int, or a function of the bucket's image
    shape (device_batch_size per bucket: a small first image must not size the launches of the large ones).  The reference walks a
    ragged region dataset one image per step (train/classif_regions.py:107-132, model/siamese.py:184); bucketing keeps the backbone
    batched whatever the mix of sizes.  Per-image results are independent, so the order of evaluation does not matter; a launch the
    device cannot hold (torch.cuda.OutOfMemoryError) is retried as two halves.
    stage = (trans, device): f is called as f(indices, items, x) with x the batch already ON the device, staged by a BatchStager per bucket --
    resident sets as row gathers, lazy / non-resident ones decoded ahead, stacked into reusable pinned buffers and copied on a copy stream while
    the previous batch computes (staging batch by batch on the compute path -- decode, clone, stack, pin, copy -- was 1.4 ms per 448 x 448
    image against 0.26 ms of kernels)."""
    buckets = {}
    for i, item in enumerate(dataset):
        buckets.setdefault((tuple(item[0].shape), item[0].dtype), []).append(i)

    def call(ii, items, x=None):
        if stage is None:
            f(ii, items)
        else:
            f(ii, items, stage_batch(items, stage[0], stage[1]) if x is None else x)

    def run(ii, items, x=None):
        try:
            call(ii, items, x)
        except torch.cuda.OutOfMemoryError:
            if len(ii) == 1:
                raise
            del x
            torch.cuda.empty_cache()
            h = len(ii) // 2
            run(ii[:h], items[:h])
            run(ii[h:], items[h:])

    for (shape, _), idx in buckets.items():
        bs = max(int(batch_size(shape) if callable(batch_size) else batch_size), 1)
        sub = dataset if len(idx) == len(dataset) else [dataset[j] for j in idx]
        stager = BatchStager(sub, bs, stage[0], stage[1]) if stage is not None else None
        for s in range(0, len(idx), bs):
            ii, items = idx[s:s + bs], sub[s:s + bs]
            run(ii, items, stager.get(s, items) if stager is not None else None)


def scatter_rows(slab, indices, rows):
    """slab[indices] = rows (consecutive index runs are plain slice copies)."""
    if indices == list(range(indices[0], indices[0] + len(indices))):
        slab[indices[0]:indices[0] + len(indices)].copy_(rows)
    else:
        slab.index_copy_(0, torch.tensor(indices, dtype=torch.int64, device=slab.device), rows.to(slab.dtype))


def load_training_sets(P, dataset_full, labels):
    """(train_set, test_train_set, test_set) of a training run from the dataset folder (or `synthetic:` spec), as the reference's training mains
    read them (train/siamese_descriptor.py:166-191): the images of `dataset_full` train AND serve as the gallery of the evaluations, those of
    `<dataset_full>/test` whose label is known are the queries.  With images pre-processed once and nothing augmented (P.train_pre_proc, the
    reference's setting: train_trans == test_trans) the training and the gallery set are the same tensors -- one list serves as both (the
    reference keeps two copies).  The dataset-dependent fields of P (reference train/*_p.py:27-48) are filled from the dataset id.  On the GPU
    the sets carry raw uint8 pixels (normalised on the device), decoded by the processes of train/_decode_farm.py."""
    from test import _common as C
    from .global_p import feature_sizes, image_sizes
    dataset_id = C.dataset_id_of(dataset_full)
    if not getattr(P, 'train_pre_proc', True) or getattr(P, 'train_trans', None) is not None:
        raise NotImplementedError("training entry point: only pre-processed, un-augmented training sets (P.train_pre_proc = True, P.train_trans = None); "
                                  "build augmented sets yourself and call main(train_set, test_train_set, test_set)")
    del labels[:]
    test_set, ref_set = C.load_sets(dataset_full, labels, raw=(P.cuda_device >= 0), lazy=False)
    P.dataset_full, P.dataset_id = dataset_full, dataset_id
    P.image_input_size = image_sizes[dataset_id]
    P.num_classes = len(labels)
    P.feature_size2d = feature_sizes[str(P.cnn_model).lower(), image_sizes[dataset_id]]
    P.test_pre_proc = True
    return ref_set, ref_set, test_set


def training_cli(argv, P, run, what):
    """`python -m train.<approach> --dataset=<folder | synthetic:...> [--model=] [--device=] [--epochs=] [--classif-model=] [--preload-net=]
    [--save-dir=] [--feature-dim=] [--batch-size=] [--micro-batch=] [--lr=]`: the reference's training scripts take everything from their
    *_p.py file (edit and run); the same fields can be given here instead."""
    import getopt
    import sys
    spec = {'dataset': ('dataset_full', str), 'model': ('cnn_model', str), 'device': ('cuda_device', int), 'epochs': ('train_epochs', int),
            'classif-model': ('classif_model', str), 'preload-net': ('preload_net', str), 'save-dir': ('save_dir', str),
            'feature-dim': ('feature_dim', int), 'batch-size': ('train_batch_size', int), 'micro-batch': ('train_micro_batch', int),
            'lr': ('train_lr', float), 'seed': ('train_seed', int)}
    try:
        opts, _ = getopt.getopt(argv, '', ['help'] + [k + '=' for k in spec])
    except getopt.GetoptError as e:
        print('%s\nusage: python -m %s %s' % (e, what, ' '.join('[--%s=]' % k for k in spec)))
        sys.exit(2)
    for opt, arg in opts:
        if opt == '--help':
            print('usage: python -m %s %s' % (what, ' '.join('[--%s=]' % k for k in spec)))
            sys.exit()
        field, typ = spec[opt[2:]]
        setattr(P, field, typ(arg))
    if not getattr(P, 'dataset_full', None):
        print('no dataset: give --dataset=<folder> (or synthetic:<dataset id>[:n=..][:q=..][:labels=..]) or set P.dataset_full')
        sys.exit(2)
    from utils.general import cap_torch_threads
    cap_torch_threads()
    # data parallel: `python -m torch.distributed.run --nproc-per-node N -m train.<approach> ...` -- one rank per GPU (RCCL; gloo for CPU runs);
    # utils.train_gen finds the process group and splits every mini-batch's micro-batches over the ranks (isx/dp.py)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get('LOCAL_RANK', '0'))
        if P.cuda_device >= 0:
            P.cuda_device = local
            torch.cuda.set_device(local)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo')
        try:
            return run()
        finally:
            dist.barrier()
            dist.destroy_process_group()
    return run()


_LABEL_IDS = {}


def label_index(labels):
    """label -> position in the run's label list (what the reference computes as `labels.index(lab)` per image -- train/classif_finetune.py:46,
    train/siamese_descriptor.py:129: O(#labels) each, seconds per pass at 10 k labels); first occurrence wins, as with list.index.  Cached per
    list object and checked against a copy of its content (the mains refill the same list object on every run)."""
    key = id(labels)
    hit = _LABEL_IDS.get(key)
    if hit is None or hit[1] != labels:          # a list compare (identical objects: pointer compares) -- the mains refill the SAME list per run
        ids = {}
        for i, lab in enumerate(labels):
            ids.setdefault(lab, i)
        _LABEL_IDS.clear()                       # one list at a time is alive in a run
        hit = _LABEL_IDS[key] = (len(labels), list(labels), ids)
    return hit[2]


def test_transform(P):
    return None if P.test_pre_proc else P.test_trans


def base_model(P, pretrained=True):
    ctor = backbones.MODELS.get(P.cnn_model.lower())
    if ctor is None:
        raise ValueError('unknown cnn_model %r' % (P.cnn_model,))
    return ctor(pretrained=pretrained)


def load_weights(net, fname):
    """net.load_state_dict(torch.load(fname)) (reference train/siamese_descriptor.py:158-160).  Files in torch's zip format are memory-mapped (the
    822 MB descriptor head is copied once, file -> parameter, not file -> RAM -> parameter); the legacy pickle files reference-era torch wrote
    are read as before."""
    if fname:
        try:
            state = torch.load(fname, map_location='cpu', mmap=True)
        except (RuntimeError, ValueError, TypeError):
            state = torch.load(fname, map_location='cpu')
        net.load_state_dict(state)
    return net


def prepare_for_inference(net, P):
    """eval mode (+ BatchNorm folding / fused epilogues when P.fold_bn): call once before get_embeddings."""
    from model.nn_utils import fold_batch_norm, set_net_train
    set_net_train(net, False)
    if getattr(P, 'fold_bn', False) and any(isinstance(m, torch.nn.BatchNorm2d) for m in net.features.modules()):
        dev = next(net.parameters()).device
        net.features = fold_batch_norm(net.features).to(dev)
        if dev.type == 'cuda':
            net.features = net.features.to(memory_format=torch.channels_last)     # MIOpen-run layers keep NHWC weights
    return net
