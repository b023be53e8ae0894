"""Shared plumbing of the four evaluation entry points: dataset loading (image folder as in
the reference, or the in-memory synthetic source), the retrieval evaluation itself and the
getopt front-end."""
from __future__ import print_function

import getopt
import os
import sys
import traceback

import torch

from train.global_p import image_sizes, match_label_functions, mean_std_files
from utils import (check_bool, check_file, check_folder, check_int, check_model, get_images_labels, log_detail,
                   parse_dataset_id, read_mean_std, retrieval_metrics, synthetic_image_set)

SYNTHETIC = 'synthetic:'


def imread_rgb(fname):
    from PIL import Image
    return Image.open(fname).convert('RGB')


def to_normalised_tensor(img, mean, std):
    """ToTensor + Normalize (reference test/*_test.py:65): HWC uint8 -> CHW float in [0,1], per-channel."""
    import numpy as np
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    m = torch.tensor(list(mean)).view(-1, 1, 1)
    s = torch.tensor(list(std)).view(-1, 1, 1)
    return (x - m) / s


def is_synthetic(dataset_full):
    return dataset_full.startswith(SYNTHETIC)


def synthetic_spec(dataset_full):
    """'synthetic:<dataset id>[:n=100][:q=20][:labels=10][:size=224][:struct=0]' -> (id, dict); struct = per cent of
    per-label pattern mixed into the noise images (utils.dataset.synthetic_image_set)."""
    parts = dataset_full[len(SYNTHETIC):].split(':')
    opts = dict(n=100, q=20, labels=10, size=224, struct=0)
    for p in parts[1:]:
        k, v = p.split('=')
        opts[k] = int(v)
    return parts[0], opts


def dataset_id_of(dataset_full):
    return synthetic_spec(dataset_full)[0] if is_synthetic(dataset_full) else parse_dataset_id(dataset_full)


def to_raw_tensor(img):
    """Decoded image as an (H,W,3) uint8 tensor: normalised later, on the GPU, by train._common.stage_batch."""
    import numpy as np
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())


class ImageLoader(object):
    """file name -> image tensor: the raw (H,W,3) uint8 pixels (raw=True, normalised later on the GPU) or the normalised (3,H,W) float tensor of
    the reference's ToTensor + Normalize.  `farm_kind` tells the decoders of train/_decode_farm.py that the file is read as plain 8-bit RGB, so
    the read may happen in a decoder process and `from_u8` finish it here."""
    farm_kind = "rgb_u8"

    def __init__(self, mean=None, std=None, raw=False):
        self.mean, self.std, self.raw = mean, std, bool(raw)

    def from_u8(self, u8):
        return u8 if self.raw else to_normalised_tensor(u8.numpy(), self.mean, self.std)

    def __call__(self, fname):
        return self.from_u8(to_raw_tensor(imread_rgb(fname)))


def load_sets(dataset_full, labels, raw=False, lazy=None):
    """(test_set, test_train_set) as lists of (normalised tensor, label, path); fills `labels`
    with the sorted label set of the reference (gallery) images; queries whose label is
    unknown are dropped (reference test/classif_finetune_test.py:62-73).  raw=True keeps folder images
    as uint8 (H,W,3) tensors and registers mean/std for the GPU-side ToTensor + Normalize.
    lazy (default: raw and ISX_LAZY_INGEST != 0): the GALLERY of a folder dataset carries train._common.LazyImage entries -- decoded batch by
    batch on a thread pool while the previous batch runs, never all in RAM; the queries (classified AND embedded: read twice) are decoded up front."""
    if is_synthetic(dataset_full):
        _, o = synthetic_spec(dataset_full)
        size = (3, o['size'], o['size'])
        ref = synthetic_image_set(o['n'], o['labels'], size, seed=1234, prefix='synthetic/ref', structure=o['struct'] / 100.0)
        qry = synthetic_image_set(o['q'], o['labels'], size, seed=4321, prefix='synthetic/test', structure=o['struct'] / 100.0)
        labels.extend(sorted(set(lab for _, lab, _ in ref)))
        return [t for t in qry if t[1] in labels], ref
    dataset_id = parse_dataset_id(dataset_full)
    match = match_label_functions[dataset_id]
    ref_files = get_images_labels(dataset_full, match)
    qry_files = get_images_labels(dataset_full + '/test', match)
    labels.extend(sorted(set(lab for _, lab in ref_files)))
    mean, std = read_mean_std(mean_std_files[dataset_id])
    if raw:
        from train import _common as TC
        TC.RAW_INGEST["mean"], TC.RAW_INGEST["std"] = list(mean), list(std)
        load = ImageLoader(raw=True)
    else:
        load = ImageLoader(mean, std)
    if lazy is None:
        lazy = raw and os.environ.get("ISX_LAZY_INGEST", "1") != "0"
    if lazy and ref_files:
        from train import _common as TC
        first = load(ref_files[0][0])
        ref = [(TC.LazyImage(f, load, first.shape, first.dtype), lab, f) for f, lab in ref_files]
    else:
        ref = [(im, lab, f) for im, (f, lab) in zip(_decode_all(load, [f for f, _ in ref_files]), ref_files)]
    qry_files = [(f, lab) for f, lab in qry_files if lab in labels]
    rank, world = dp_world()
    if world > 1 and raw and qry_files:
        # data-parallel evaluation: a rank classifies and embeds only ITS contiguous slice of the queries (dp_classify / dp_embeddings) -- only
        # those files are decoded here; the others keep their place in the list (labels and paths are what the metrics read of them)
        from train import _common as TC
        lo, hi = dp_bounds(len(qry_files), world, rank)
        mine = _decode_all(load, [f for f, _ in qry_files[lo:hi]])
        first = mine[0] if mine else load(qry_files[0][0])
        qry = [(mine[i - lo] if lo <= i < hi else TC.LazyImage(f, load, first.shape, first.dtype), lab, f) for i, (f, lab) in enumerate(qry_files)]
        return qry, ref
    qry = [(im, lab, f) for im, (f, lab) in zip(_decode_all(load, [f for f, _ in qry_files]), qry_files)]
    return qry, ref


FARM_MIN_FILES = int(os.environ.get("ISX_DECODE_FARM_MIN", "64"))


def _decode_all(load, files):
    """[load(f) for f in files], file order kept.  The reference reads its folders one image at a time on the main thread
    (test/classif_finetune_test.py:62-73), which is what a GPU test run then waits for; here the files of an ImageLoader go to the decoder
    processes of train/_decode_farm.py (a thread pool does not scale: the decode of a small JPEG is mostly Python under the GIL) and only the
    float conversion of the non-raw form runs on threads.  ISX_DECODE_PROCS=0 keeps everything on threads, ISX_DECODE_THREADS=1 sequential."""
    files = list(files)
    workers = int(os.environ.get("ISX_DECODE_THREADS", "0")) or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)
    if workers <= 1 or len(files) < 2 * workers:
        return [load(f) for f in files]
    from concurrent.futures import ThreadPoolExecutor
    if getattr(load, "farm_kind", None) == "rgb_u8" and len(files) >= FARM_MIN_FILES:
        from train import _decode_farm as DF
        if DF.decode_farm() is not None:
            first = to_raw_tensor(imread_rgb(files[0]))
            slot = max(1 << 20, 2 * first.numel())            # ragged folders: a file larger than the slot is decoded in this process (Ticket.tensor)
            u8 = DF.decode_files(files, slot)
            if load.raw:
                return u8
            with ThreadPoolExecutor(max_workers=workers) as pool:
                return list(pool.map(load.from_u8, u8))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        return list(pool.map(load, files))


# ---- data-parallel evaluation: `python -m torch.distributed.run --nproc-per-node N -m test.<approach>_test ...` ------------------------
# Extraction is embarrassingly parallel (images are independent): every rank extracts a contiguous 1/N of the queries and of the gallery, the
# descriptor rows are all-gathered (1 M x 2048 fp32 = 8.2 GB: every GPU of the node holds the whole slab with room to spare), and the metrics are
# split by QUERY rows -- P@1 and AP of a query need its whole score row, which each rank computes against the full gallery for its queries, with
# the same kernels on the same values as one process would.  The counts are summed, the per-query APs gathered in query order and averaged as
# the single-process run averages them: the printed lines are the same.  (Galleries beyond one GPU's memory: isx.retrieval.ShardedGallery,
# top-k metrics -- bench.py's retrieval_shard leg.)
def dp_world():
    import torch.distributed as dist
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def dp_bounds(n, world, rank):
    """[lo, hi): the contiguous slice of n items that belongs to `rank`."""
    return (n * rank) // world, (n * (rank + 1)) // world


def _dp_gather_rows(local, counts, device):
    """All ranks' row blocks (rank r holds counts[r] rows) -> the whole (sum(counts), D) tensor on every rank, rows in rank order."""
    import torch.distributed as dist
    world, D, most = len(counts), local.size(1), max(counts)
    pad = local.new_zeros((most, D))
    pad[:local.size(0)].copy_(local)
    if dist.get_backend() == 'nccl':
        out = pad.new_empty((world * most, D))
        dist.all_gather_into_tensor(out, pad)
        parts = [out[r * most:r * most + counts[r]] for r in range(world)]
    else:
        host = pad.cpu()
        bufs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(bufs, host)
        parts = [bufs[r][:counts[r]].to(local.device) for r in range(world)]
    return torch.cat(parts, 0)


def dp_sync_net(net):
    """Every rank evaluates rank 0's network: without a weights file the layers a wrapper adds (a new classifier, the descriptor head) are
    random-initialised, differently on every rank."""
    if dp_world()[1] > 1:
        from isx.dp import broadcast_module_state
        broadcast_module_state(net, src=0)
    return net


def dp_classify(test_classif_net, net, test_set):
    """test_classif_net over the ranks: (correct, total) of the whole set on every rank."""
    rank, world = dp_world()
    if world == 1:
        return test_classif_net(net, test_set)
    import torch.distributed as dist
    lo, hi = dp_bounds(len(test_set), world, rank)
    c, t = test_classif_net(net, test_set[lo:hi]) if hi > lo else (0, 0)
    dev = next(net.parameters()).device if dist.get_backend() == 'nccl' else 'cpu'
    tot = torch.tensor([int(c), int(t)], dtype=torch.int64, device=dev)
    dist.all_reduce(tot)
    return int(tot[0]), int(tot[1])


def dp_embeddings(get_embeddings, net, dataset, device, out_size):
    """get_embeddings over the ranks: each extracts its contiguous slice of `dataset`, every rank returns the whole (len(dataset), out_size) slab."""
    rank, world = dp_world()
    if world == 1:
        return get_embeddings(net, dataset, device, out_size)
    counts = [dp_bounds(len(dataset), world, r)[1] - dp_bounds(len(dataset), world, r)[0] for r in range(world)]
    lo, hi = dp_bounds(len(dataset), world, rank)
    if hi > lo:
        local = get_embeddings(net, dataset[lo:hi], device, out_size)
    else:
        local = torch.zeros((0, out_size), device=('cuda:%d' % device if device >= 0 else 'cpu'))
    return _dp_gather_rows(local, counts, device)


def dp_metrics(test_embeddings, ref_embeddings, test_set, ref_set):
    """retrieval_metrics with the QUERIES split over the ranks (each against the whole gallery); the dict of the whole query set on every rank."""
    rank, world = dp_world()
    if world == 1:
        return retrieval_metrics(test_embeddings, ref_embeddings, test_set, ref_set)
    import torch.distributed as dist
    lo, hi = dp_bounds(len(test_set), world, rank)
    if hi > lo:
        m = retrieval_metrics(test_embeddings[lo:hi], ref_embeddings, test_set[lo:hi], ref_set)
        mine = (int(m['correct']), list(m['aps']))
    else:
        mine = (0, [])
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    correct = sum(c for c, _ in parts)
    aps = [a for _, ap in parts for a in ap]                # rank order = query order
    valid = [a for a in aps if a == a]
    M = len(test_set)
    return {'prec1': float(correct) / M, 'correct': correct, 'total': M, 'mAP': sum(valid) / float(len(valid)) if valid else float('nan'), 'aps': aps}


class GalleryShard(object):
    """This rank's contiguous rows of a gallery that is NOT gathered (ISX_EVAL_SHARDED=1, or a slab beyond SHARD_GALLERY_BYTES): `rows` are the
    descriptors of gallery items [lo, lo + len(rows)) of `total`."""

    def __init__(self, rows, lo, total):
        self.rows, self.lo, self.total = rows, int(lo), int(total)

    def size(self, dim=None):
        shape = (self.total, self.rows.size(1))
        return shape if dim is None else shape[dim]


SHARD_GALLERY_BYTES = 64 << 30           # a gallery slab beyond this stays sharded over the ranks (one MI355X holds 288 GB: 1 M x 2048 fp32 is 8.2 GB)


def _keep_sharded(n_rows, out_size):
    mode = os.environ.get('ISX_EVAL_SHARDED', 'auto')
    if dp_world()[1] == 1 or mode == '0':
        return False
    return mode == '1' or n_rows * out_size * 4 > SHARD_GALLERY_BYTES


def dp_metrics_sharded(test_embeddings, shard, test_set, ref_set, kth=1):
    """P@1 and mAP against a gallery that stays sharded by rows: the kth-best gallery item of every query through the sharded search (per-shard
    top-k, all-gather, canonical merge: isx.retrieval.ShardedGallery), the average precisions through the sharded rank counts
    (utils.metrics.sharded_average_precisions) -- the values of the unsharded evaluation, no rank ever holds the whole slab or a whole score row."""
    from isx.retrieval import ShardedGallery
    from utils.metrics import _label_ids, sharded_average_precisions
    qlab, glab = _label_ids(test_set, ref_set)
    M = len(test_set)
    _, idx = ShardedGallery(shard.rows, idx_base=shard.lo).search(test_embeddings, kth)
    best = idx[:, kth - 1].cpu().tolist()
    correct = sum(1 for i, j in enumerate(best) if ref_set[j][1] == test_set[i][1])
    aps = sharded_average_precisions(test_embeddings, shard.rows, shard.lo, qlab, glab[shard.lo:shard.lo + shard.rows.size(0)], kth).tolist()
    if any(a == -1.0 for a in aps):
        raise NotImplementedError('a query has more than 2048 positives: the sharded average precision does not cover it (gather the gallery: ISX_EVAL_SHARDED=0)')
    valid = [a for a in aps if a == a]
    return {'prec1': float(correct) / M, 'correct': correct, 'total': M, 'mAP': sum(valid) / float(len(valid)) if valid else float('nan'), 'aps': aps}


def gallery_embeddings(get_embeddings, net, ref_set, device, out_size, labels, save_slab=None, gallery_slab=None):
    """(gallery descriptors, gallery set) of an evaluation run.  Default: extracted, as the reference does on every run
    (test/classif_finetune_test.py:80-81).  --save-slab=<file>: the extracted slab is also written to disk (isx/slab.py: row blocks streamed from
    the GPU, labels, + <file>.labels.json); --gallery-slab=<file>: nothing is extracted -- the slab is read back (mmap -> pinned staging -> HBM)
    and the run ranks against it; the label set of the file must be the run's."""
    from isx import slab
    if gallery_slab:
        dev = 'cuda:%d' % device if device >= 0 else 'cpu'
        desc, slab_set, names = slab.load_gallery(gallery_slab, dev)
        if list(names) != list(labels):
            raise ValueError('--gallery-slab: %s was written for another label set (%d labels, this run has %d)' % (gallery_slab, len(names), len(labels)))
        if desc.size(1) != out_size:
            raise ValueError('--gallery-slab: %s holds %d-d descriptors, this run computes %d-d ones' % (gallery_slab, desc.size(1), out_size))
        print('Gallery: {0} descriptors read from {1}'.format(desc.size(0), gallery_slab))
        return desc, slab_set
    if _keep_sharded(len(ref_set), out_size) and not save_slab:
        rank, world = dp_world()
        lo, hi = dp_bounds(len(ref_set), world, rank)
        local = get_embeddings(net, ref_set[lo:hi], device, out_size) if hi > lo else torch.zeros((0, out_size), device=('cuda:%d' % device if device >= 0 else 'cpu'))
        return GalleryShard(local, lo, len(ref_set)), ref_set
    emb = dp_embeddings(get_embeddings, net, ref_set, device, out_size)
    if save_slab:
        if dp_world()[0] == 0:                               # every rank holds the whole slab; one writes it
            slab.save_gallery(save_slab, emb, ref_set, labels)
        print('Gallery: {0} descriptors written to {1}'.format(emb.size(0), save_slab))
    return emb, ref_set


def evaluate_retrieval(test_embeddings, ref_embeddings, test_set, ref_set, device, labels, dba):
    """sim -> P@1 + mAP, the reference's result line, then the optional DBA pass.  Returns the
    (prec1, mAP) pair of the plain pass (the reference returns nothing; the prints are the API)."""
    # sim = torch.mm(test_emb, ref_emb.t()) -> precision1 -> mean_avg_precision (reference test/classif_finetune_test.py:82-85),
    # evaluated in query-row blocks when the matrix exceeds utils.metrics.SIM_BUDGET_BYTES (same values, no 40 GB matrix)
    if isinstance(ref_embeddings, GalleryShard):
        if dba != 0:
            raise NotImplementedError('DBA needs the whole gallery on one device: run with ISX_EVAL_SHARDED=0')
        m = dp_metrics_sharded(test_embeddings, ref_embeddings, test_set, ref_set)
    else:
        m = dp_metrics(test_embeddings, ref_embeddings, test_set, ref_set)
    prec1, mAP = m['prec1'], m['mAP']
    print('Descriptor (TEST): {0} / {1} - acc: {2:.4f} - mAP:{3:.4f}'.format(m['correct'], m['total'], prec1, mAP))
    if dba == 0:
        return prec1, mAP
    from .instance_avg import instance_avg
    print('Testing using instance feature augmentation')
    dba_embeddings, dba_set = instance_avg(device, ref_embeddings, ref_set, labels, dba)
    d = dp_metrics(test_embeddings, dba_embeddings, test_set, dba_set)
    print('Descriptor (TEST DBA k={4}): {0} / {1} - acc: {2:.4f} - mAP:{3:.4f}'.format(d['correct'], d['total'], d['prec1'], d['mAP'], dba))
    return prec1, mAP


def run_cli(argv, usage, spec, required, main, P):
    """getopt front-end shared by the four scripts.  `spec`: option name -> (kind, label) with
    kind in {'dataset','model','file','int','bool'}; returns after calling main(**values)."""
    try:
        opts, _ = getopt.getopt(argv, '', ['help', 'sharded='] + [name + '=' for name in spec])
    except getopt.GetoptError:
        usage()
        sys.exit(2)
    values = dict((name.replace('-', '_'), None) for name in spec)
    values['dba'] = -1 if 'dba' in spec else None
    for opt, arg in opts:
        if opt == '--help':
            usage()
            sys.exit()
        if opt == '--sharded':                           # every main: keep the gallery sharded over the ranks of a torch.distributed.run launch
            os.environ['ISX_EVAL_SHARDED'] = '1' if check_bool(arg, 'sharded', usage) else '0'
            continue
        name = opt[2:]
        kind, label = spec[name]
        if kind == 'dataset':
            val = arg if is_synthetic(arg) else check_folder(arg, label, True, usage)
        elif kind == 'model':
            val = check_model(arg, usage)
        elif kind == 'file':
            val = check_file(arg, label, True, usage)
        elif kind == 'path':
            val = arg
        elif kind == 'int':
            val = check_int(arg, label, usage)
        else:
            val = check_bool(arg, label, usage)
        values[name.replace('-', '_')] = val
    if any(values[r] is None for r in required):
        print('One or more required arguments is missing.')
        usage()
        sys.exit(2)
    if 'dba' not in spec:
        values.pop('dba', None)
    from utils.general import cap_torch_threads
    cap_torch_threads()
    device = values['device']
    # data parallel: under `python -m torch.distributed.run --nproc-per-node N` every rank takes GPU LOCAL_RANK (RCCL; gloo for CPU runs, and with
    # ISX_BENCH_ONE_DEVICE=1 -- a rehearsal on a one-GPU box: every rank on cuda:0); rank 0 prints
    world = int(os.environ.get('WORLD_SIZE', '1'))
    quiet = None
    if world > 1:
        import torch.distributed as dist
        one_device = os.environ.get('ISX_BENCH_ONE_DEVICE', '0') == '1'
        if device >= 0 and not one_device:
            device = values['device'] = int(os.environ.get('LOCAL_RANK', '0'))
            torch.cuda.set_device(device)
            dist.init_process_group('nccl', device_id=torch.device('cuda', device))
        else:
            dist.init_process_group('gloo')
        if dist.get_rank() != 0:
            quiet = open(os.devnull, 'w')
    try:
        import contextlib
        with (contextlib.redirect_stdout(quiet) if quiet is not None else contextlib.nullcontext()):
            if device >= 0:
                with torch.cuda.device(device):
                    return main(**values)
            else:
                return main(**values)
    except Exception:
        log_detail(P, None, traceback.format_exc())
        raise
    finally:
        if world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()
        if quiet is not None:
            quiet.close()


def usage_text(script, lines):
    print('Usage: ' + script + ' [options]')
    print('Options:\n\tRequired:\n' + ''.join(lines) + '--help\t\tShow this help\n')


O_DATASET = ('--dataset=\t<path>\tThe path to the dataset containing all reference images. It should contain a '
             'sub-folder "test" containing all test images (or synthetic:<id>[:n=..][:q=..][:labels=..])\n')
O_MODEL = '--model=\t<name>\tAlexNet, ResNet152 or ResNet50 to specify the type of model.\n'
O_DEVICE = '--device=\t<int>\tThe GPU device used for testing. If negative, CPU is used.\n'
O_DBA = ('--dba=\t<int>\tUse DBA with given k. If k = 0, do not use DBA. If k<0, use all neighbors within the '
         'same instance.\n')
O_SLAB = ('--save-slab=\t<file>\t(extension) Also write the gallery descriptors to this slab file.\n'
          '--gallery-slab=\t<file>\t(extension) Read the gallery descriptors from this slab file instead of extracting them.\n'
          '--sharded=\t<bool>\t(extension, under torch.distributed.run) Keep the gallery sharded by rows over the ranks instead of gathering it '
          '(sharded search + sharded average precision; default: only beyond 64 GB).\n')
O_BATCH = '--batch=\t<int>\tThe batch size to use.\n'
