"""Average precision against a gallery sharded by rows (reference utils/metrics.py:25-45 ranks one full score row): the three steps
of include/isx.h (isx_ap_shard_*) give, for any number of shards, the float64 bits of the unsharded evaluation."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(M, N, n_labels, seed, ties=True):
    g = torch.Generator().manual_seed(seed)
    sim = torch.randn(M, N, generator=g)
    if ties:
        sim = (sim * 4).round() / 4                          # many exactly equal scores: the index tie-break decides
        sim[:, N // 3] = sim[:, N // 3 + 1]
        sim[0, :5] = 0.0
        sim[0, 2] = -0.0                                     # -0.0 ranks as +0.0
    glab = torch.randint(0, n_labels, (N,), generator=g, dtype=torch.int32)
    qlab = torch.randint(0, n_labels + 1, (M,), generator=g, dtype=torch.int32)      # label n_labels: a query without positives
    return sim, qlab, glab


def _sharded(sim, qlab, glab, bounds, kth, device="cpu"):
    """The three steps with the collectives replaced by cat / sum over the shards, all in this process."""
    from utils import metrics as MT
    keys, counts, hists = [], [], []
    for lo, hi in bounds:
        k, c = MT.ap_shard_positives(sim[:, lo:hi].contiguous().to(device), lo, qlab.to(device), glab[lo:hi].contiguous().to(device))
        keys.append(k.cpu()); counts.append(c.cpu())
    keys_all = torch.cat(keys, 1)
    n_lab = torch.stack(counts, 0).sum(0).to(torch.int32)
    for lo, hi in bounds:
        hists.append(MT.ap_shard_hist(sim[:, lo:hi].contiguous().to(device), lo, keys_all.to(device)).cpu())
    hist = torch.stack(hists, 0).sum(0).to(torch.int32)
    return MT.ap_from_hist(hist.to(device), n_lab.to(device), kth).cpu(), keys_all, hist, n_lab


def _same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)) and torch.equal(a.isnan(), b.isnan())


@pytest.mark.parametrize("bounds", [[(0, 97)], [(0, 30), (30, 31), (31, 97)], [(0, 50), (50, 50), (50, 97)]])
@pytest.mark.parametrize("kth", [1, 2])
def test_cpu_restatement_equals_the_unsharded_evaluation(bounds, kth):
    from utils import metrics as MT
    sim, qlab, glab = _case(14, 97, 6, 3)
    want = MT._average_precisions(sim, qlab, glab, kth)
    got, _, _, n_lab = _sharded(sim, qlab, glab, bounds, kth)
    assert _same(got, want) and want.isnan().any() and int(n_lab.max()) > 5
    # and the oracle's restatement of the reference loop on the canonically ranked list (oracle/isx_oracle.c, pinned by the reference's own function)
    import oracle as O
    ora = torch.from_numpy(O.average_precision(O.rank_full(sim.numpy()), qlab.numpy(), glab.numpy(), kth))
    assert _same(got, ora)


def _two_ranks(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from utils import metrics as MT
    g = torch.Generator().manual_seed(0)
    G = torch.nn.functional.normalize(torch.randn(83, 16, generator=g), dim=1)
    G[40] = G[7]                                             # a tie across the shard boundary
    Q = torch.nn.functional.normalize(torch.randn(9, 16, generator=g), dim=1)
    glab = torch.randint(0, 5, (83,), generator=g, dtype=torch.int32)
    qlab = torch.randint(0, 5, (9,), generator=g, dtype=torch.int32)
    lo, hi = (83 * rank) // world, (83 * (rank + 1)) // world
    ap = MT.sharded_average_precisions(Q, G[lo:hi], lo, qlab, glab[lo:hi], budget_bytes=4 * 4 * 83)      # several query blocks
    if rank == 0:
        torch.save((ap, MT._average_precisions(Q @ G.t(), qlab, glab, 1)), out)
    dist.barrier()
    dist.destroy_process_group()


def _eight_ranks(rank, world, port, out):
    """Eight gloo ranks, rank 3 holding an EMPTY shard (N = 7 rows over ranks 0-2 and 4-7): sharded search + exchange + merge, and the sharded
    average precision, against the unsharded evaluation."""
    sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isx import retrieval as R
    from utils import metrics as MT
    g = torch.Generator().manual_seed(1)
    N = 203
    G = torch.nn.functional.normalize(torch.randn(N, 16, generator=g), dim=1)
    G[150] = G[7]                                            # a tie across shards
    Q = torch.nn.functional.normalize(torch.randn(11, 16, generator=g), dim=1)
    glab = torch.randint(0, 6, (N,), generator=g, dtype=torch.int32)
    qlab = torch.randint(0, 7, (11,), generator=g, dtype=torch.int32)
    cuts = [0, 40, 41, 90, 90, 120, 160, 161, N]             # rank 3: rows [90, 90)
    lo, hi = cuts[rank], cuts[rank + 1]
    ap = MT.sharded_average_precisions(Q, G[lo:hi], lo, qlab, glab[lo:hi])
    s, i = R.ShardedGallery(G[lo:hi], lo).search(Q, 10)
    if rank == 0:
        us, ui = R.local_topk(Q, G, 10, 0)
        torch.save((ap, MT._average_precisions(Q @ G.t(), qlab, glab, 1), s, i, us, ui), out)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_gloo_ranks_one_of_them_empty_handed(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "ap8.pt")
    mp.spawn(_eight_ranks, args=(8, port, out), nprocs=8, join=True)
    ap, want, s_, i_, us, ui = torch.load(out)
    assert ap.shape == (11,) and _same(ap, want)
    assert torch.equal(i_, ui) and torch.allclose(s_, us, rtol=0, atol=1e-6)


def test_two_gloo_ranks_compute_the_unsharded_average_precisions(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "ap.pt")
    mp.spawn(_two_ranks, args=(2, port, out), nprocs=2, join=True)
    got, want = torch.load(out)
    assert got.shape == (9,) and _same(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,n_labels,bounds", [
    (14, 97, 6, [(0, 97)]),
    (14, 97, 6, [(0, 30), (30, 31), (31, 97)]),
    (33, 4096, 40, [(0, 1024), (1024, 4096)]),
    (8, 200000, 2000, [(0, 50000), (50000, 120004), (120004, 200000)]),
    # eight shards (BASELINE configs[4]'s rank count), one of them EMPTY, one a single row, boundaries off the 16-byte grid
    (33, 40007, 97, [(0, 5001), (5001, 5001), (5001, 5002), (5002, 15000), (15000, 20003), (20003, 29999), (29999, 35000), (35000, 40007)]),
])
@pytest.mark.parametrize("kth", [1, 2])
def test_kernels_equal_the_unsharded_kernel_and_the_cpu_restatement(M, N, n_labels, bounds, kth):
    """isx_ap_shard_positives / _hist / isx_ap_from_hist on the GPU: the same key sets, the same histograms and the same float64 APs as the
    CPU restatement, and the APs of isx_average_precision_sim on the whole matrix, bit for bit -- ties, -0.0, queries without positives, shard
    boundaries off the 16-byte grid, an empty-handed shard."""
    from isx import ops
    sim, qlab, glab = _case(M, N, n_labels, 11 + N, ties=N < 10000)
    got, keys_all, hist, n_lab = _sharded(sim, qlab, glab, bounds, kth, device="cuda")
    ref, keys_cpu, hist_cpu, n_lab_cpu = _sharded(sim, qlab, glab, bounds, kth, device="cpu")
    assert torch.equal(n_lab, n_lab_cpu) and torch.equal(hist, hist_cpu)
    assert torch.equal(keys_all.sort(1).values, keys_cpu.sort(1).values)                  # the same keys (the slot order is arbitrary)
    want = ops.average_precision_sim(sim.cuda(), qlab.cuda(), glab.cuda(), kth).cpu()
    assert _same(got, want) and _same(ref, want)


@pytest.mark.gpu
def test_sharded_gallery_average_precisions_one_rank():
    """utils.metrics.sharded_average_precisions without a process group (one shard = the whole gallery), query rows in several blocks."""
    from isx import ops
    from utils import metrics as MT
    g = torch.Generator().manual_seed(5)
    G = ops.l2norm_rows(torch.randn(5000, 64, generator=g).cuda())
    Q = ops.l2norm_rows(torch.randn(300, 64, generator=g).cuda())
    glab = torch.randint(0, 50, (5000,), generator=g, dtype=torch.int32)
    qlab = torch.randint(0, 50, (300,), generator=g, dtype=torch.int32)
    got = MT.sharded_average_precisions(Q, G, 0, qlab, glab, budget_bytes=128 * 5000 * 4)
    want = ops.average_precision_sim(ops.cosine_sim(Q, G), qlab.cuda(), glab.cuda(), 1).cpu()
    assert _same(got, want)
