"""Siamese triplet training step: losses, negative mining, couple generation, the
gradient-accumulating training loop and its data-parallel (gloo, world_size 2) execution."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_losses_match_reference_forward(golden):
    from model.custom_modules import MetricLoss, TripletLoss, TripletLossFun
    g = golden("training.npz")
    a, p, n = t(g["a"]), t(g["p"]), t(g["n"])
    for normalized in (True, False):
        for avg in (True, False):
            want = float(g["loss_n%d_a%d" % (normalized, avg)][0])
            got = TripletLoss(0.1, avg, normalized)(a, p, n)
            assert got.shape == (1,) and abs(float(got) - want) <= 1e-6
            assert abs(float(TripletLossFun(0.1, avg, normalized)(a, p, n)) - want) <= 1e-6
            lo, rows, ga, gp, gn = O.triplet_loss(g["a"], g["p"], g["n"], 0.1, normalized, avg)
            assert abs(lo - want) <= 1e-6 and (rows == 0).any() and (rows > 0).any()
            # analytic backward == oracle gradients
            ar, pr, nr = (x.clone().requires_grad_(True) for x in (a, p, n))
            TripletLoss(0.1, avg, normalized)(ar, pr, nr).backward()
            np.testing.assert_allclose(ar.grad.numpy(), ga, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(pr.grad.numpy(), gp, rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(nr.grad.numpy(), gn, rtol=1e-6, atol=1e-7)
    assert abs(float(MetricLoss(True)(a, p, t(g["metric_y"]))) - float(g["metric_loss"][0])) <= 1e-5
    # backward == derivative of the forward formula (double precision autograd as the judge)
    ad, pd_, nd = (x.double().clone().requires_grad_(True) for x in (a, p, n))
    l = ((ad * nd).sum(1) - (ad * pd_).sum(1) + 0.1).clamp(min=0).sum()
    l.backward()
    ar, pr, nr = (x.clone().requires_grad_(True) for x in (a, p, n))
    TripletLoss(0.1, False, True)(ar, pr, nr).backward()
    np.testing.assert_allclose(ar.grad.numpy(), ad.grad.float().numpy(), rtol=1e-6, atol=1e-7)


def test_mining_cpu_path_matches_reference(golden):
    from train.siamese_descriptor import mine_epoch_negatives
    g = golden("training.npz")
    ds = [(None, int(l), None) for l in g["mine_labels"]]
    couples = [(int(g["mine_labels"][a]), (int(a), int(b)), (None, None)) for a, b in zip(g["mine_i1"], g["mine_i2"])]
    for semi in (1, 0):
        got = mine_epoch_negatives(t(g["mine_sim"]), ds, couples, bool(semi))
        np.testing.assert_array_equal(got.numpy(), g["neg_semi%d" % semi])
        np.testing.assert_array_equal(O.mine_negatives(g["mine_sim"], g["mine_labels"], g["mine_i1"], g["mine_i2"], semi), g["neg_semi%d" % semi])


def test_pos_couples_and_shuffle():
    from utils import choose_rand_neg, get_pos_couples
    from train.siamese_descriptor import shuffle_couples
    ds = [("x%d" % i, "abcab"[i], None) for i in range(5)]
    c = get_pos_couples(ds)
    assert list(c) == ["a", "b", "c"]
    assert [x[1] for x in c["a"]] == [(0, 0), (0, 3), (3, 3)] and [x[1] for x in c["c"]] == [(2, 2)]
    assert c["a"][1][2] == ("x0", "x3")
    assert [x[1] for x in get_pos_couples(ds, duplicate=False)["b"]] == [(1, 4)]
    random.seed(3)
    big = get_pos_couples([(i, i % 7, None) for i in range(70)])
    out = shuffle_couples(big)
    assert sorted(x[1] for x in out) == sorted(x[1] for v in big.values() for x in v)       # a permutation
    assert len(set(x[0] for x in out[:7])) == 7                                              # first round covers all labels
    random.seed(0)
    assert choose_rand_neg([(0, "a", None), (1, "b", None)], "a") == 1


class _TinyNet(nn.Module):
    """Stands in for DescriptorNet in the loop tests: features + a descriptor head, 3 inputs in training."""

    def __init__(self, head_out=8):
        super().__init__()
        from model.custom_modules import RowDeferredLinear
        self.features = nn.Sequential(nn.Conv2d(3, 4, 3, stride=2), nn.BatchNorm2d(4), nn.ReLU())
        self.head = RowDeferredLinear(4 * 3 * 3, head_out)   # as DescriptorNet's head: weight gradient from the step's (x, dy) rows
        self.feature_size = head_out

    def one(self, x):
        y = self.head(self.features(x).flatten(1))
        return y / (y.pow(2).sum(1, keepdim=True) + 1e-10).sqrt()

    def forward(self, a, p=None, n=None):
        return (self.one(a), self.one(p), self.one(n)) if self.training and n is not None else self.one(a)


def _run_training(rank, world, port, out, train_bn=False, save_all=False, mode="tree", batch=8, micro=2, head_out=8, stats_out=None):
    sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
    torch.set_num_threads(1)                # same CPU kernels (and summation order) whatever the number of processes
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from model.custom_modules import TripletLoss
    from train import siamese_descriptor as sd
    from utils import train_gen
    import torch.optim as optim
    torch.manual_seed(1000 * rank)          # replicas start from DIFFERENT weights: train_gen must broadcast rank 0's
    random.seed(0)
    net = _TinyNet(head_out)
    P = sd.P
    P.cuda_device, P.train_epochs, P.train_batch_size, P.train_micro_batch = -1, 2, batch, micro
    P.train_grad_exchange = mode
    P.train_loss_int, P.train_test_int, P.test_batch_size, P.feature_dim, P.train_seed = 1000, 1000, 8, head_out, 5
    P.train_epoch_switch, P.train_pre_proc, P.train_loss_avg = 1, True, False
    P.train_bn = bool(train_bn)
    g = torch.Generator().manual_seed(1)
    ds = [(torch.randn(3, 8, 8, generator=g), "l%d" % (i % 4), "p%d" % i) for i in range(16)]
    del sd.labels[:]
    sd.labels.extend(sorted(set(l for _, l, _ in ds)))
    opt = optim.SGD(net.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    sd.test_print_descriptor = lambda *a, **k: 0            # no evaluation inside this test
    if world == 1:
        # train_gen re-seeds `random` per epoch only under DP (identical couple order on all ranks);
        # the single-process run mimics that seeding so both runs see the same triplets
        import utils.train_general as tg
        real_train_gen = tg.train_gen

        def seeded_train_gen(*a, **k):
            create_epoch = a[8]

            def ce(epoch, train_set, testset_tuple):
                random.seed(P.train_seed + epoch)
                return create_epoch(epoch, train_set, testset_tuple)
            a2 = list(a)
            a2[8] = ce
            return real_train_gen(*a2, **k)
        sd.train_gen = seeded_train_gen
    sd.train_siam_triplets_pos_couples(net, ds, (ds[:4], ds), TripletLoss(P.triplet_margin, False), opt)
    if stats_out and rank == 0:
        from isx import dp as _dp
        torch.save(dict(_dp.STATS), stats_out)
    if rank == 0 or save_all:
        torch.save({k: v.clone() for k, v in net.state_dict().items()}, out + (".%d" % rank if save_all else ""))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _moved(a):
    torch.manual_seed(0)
    init = _TinyNet().state_dict()
    return sum(float((a[k].float() - init[k].float()).abs().sum()) for k in a), init


@pytest.mark.parametrize("world,batch,micro", [(2, 8, 2), (4, 8, 2), (2, 6, 2), (4, 4, 2)])
def test_data_parallel_training_is_bit_identical_to_single_process(tmp_path, world, batch, micro):
    """P ranks x the micro-batches of their subtree + TreeExchange + row-deferred head gradient == 1 process with the whole mini-batch,
    BIT FOR BIT on the whole state dict after 2 epochs of SGD with momentum (gradient accumulation in the canonical tree order,
    BatchNorm frozen).  (2, 6, 2): 3 leaves on 2 ranks, an uneven tree; (4, 4, 2): more ranks than leaves, two ranks idle."""
    single = str(tmp_path / "single.pt")
    dp = str(tmp_path / "dp.pt")
    mp.spawn(_run_training, args=(1, 0, single, False, False, "tree", batch, micro), nprocs=1, join=True)
    mp.spawn(_run_training, args=(world, _free_port(), dp, False, True, "tree", batch, micro), nprocs=world, join=True)
    a = torch.load(single)
    moved, init = _moved(a)
    assert moved > 1e-3                                          # training really changed the weights
    assert torch.equal(a["features.1.running_mean"], init["features.1.running_mean"])     # BN frozen (train_bn False)
    for r in range(world):
        b = torch.load(dp + ".%d" % r)
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k], b[k]), (r, k, float((a[k].float() - b[k].float()).abs().max()))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_head_training_is_bit_identical_to_single_process(tmp_path, world):
    """The descriptor head sharded by OUTPUT FEATURES across the ranks (isx/shard_head.py; reference model/siamese.py:104-114 is one replicated
    822 MB weight): 8 micro-batches on 2 / 4 / 8 ranks, every rank computing its feature groups' slice of the head for the rows of ALL ranks,
    forming and applying the update of ITS rows of the weight only, the input gradient assembled from the ranks' per-group chains in group
    order -- and the whole state dict of every rank after 2 epochs of SGD with momentum is BIT FOR BIT the single process's.  The stale rows
    of the other ranks are refreshed by the sync before each epoch's embedding pass and at the end; the exchange counters say the head moved
    rows, not a weight gradient."""
    single, dpf, stats = str(tmp_path / "single.pt"), str(tmp_path / "dp.pt"), str(tmp_path / "stats.pt")
    mp.spawn(_run_training, args=(1, 0, single, False, False, "tree", 16, 2, 256), nprocs=1, join=True)
    mp.spawn(_run_training, args=(world, _free_port(), dpf, False, True, "tree", 16, 2, 256, stats), nprocs=world, join=True)
    a = torch.load(single)
    assert _moved_from(a, 256) > 1e-3
    for r in range(world):
        b = torch.load(dpf + ".%d" % r)
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k], b[k]), (r, k, float((a[k].float() - b[k].float()).abs().max()))
    st = torch.load(stats)
    assert st.get("head_shard_bytes_received", 0) > 0 and "rows_all_gather_bytes_received" not in st


def _moved_from(a, head_out):
    torch.manual_seed(0)
    init = _TinyNet(head_out).state_dict()
    return sum(float((a[k].float() - init[k].float()).abs().sum()) for k in a)


def test_data_parallel_allreduce_mode_matches_single_process(tmp_path):
    """P.train_grad_exchange = "allreduce": 2 ranks x half of every mini-batch + bucketed gradient all-reduce (+ the row-deferred head
    gradient) == 1 process, up to the rounding of a different summation order."""
    single = str(tmp_path / "single.pt")
    dp = str(tmp_path / "dp.pt")
    mp.spawn(_run_training, args=(1, 0, single, False, False, "allreduce"), nprocs=1, join=True)
    mp.spawn(_run_training, args=(2, _free_port(), dp, False, False, "allreduce"), nprocs=2, join=True)
    a, b = torch.load(single), torch.load(dp)
    assert set(a) == set(b)
    for k in a:
        np.testing.assert_allclose(a[k].numpy(), b[k].numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    assert _moved(a)[0] > 1e-3


def test_data_parallel_train_bn_keeps_replicas_identical(tmp_path):
    """P.train_bn under data parallelism (round-2 ADVICE): every rank updates its BatchNorm running statistics from its own slice of
    the mini-batch; they are averaged after every optimizer step, so the replicas remain ONE model -- identical state_dict on both
    ranks (weights AND running statistics), statistics that really moved."""
    out = str(tmp_path / "bn.pt")
    mp.spawn(_run_training, args=(2, _free_port(), out, True, True), nprocs=2, join=True)
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    torch.manual_seed(0)
    init = _TinyNet().state_dict()
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert not torch.equal(a["features.1.running_mean"], init["features.1.running_mean"])
    assert not torch.equal(a["features.1.running_var"], init["features.1.running_var"])


def _run_uneven_reducer(rank, world, port, out):
    """Rank 0 arms and runs a backward, rank 1 has an empty slice (never arms, no backward); a second step detaches the
    gradients with set_to_none before the backward.  Every rank must issue the same per-bucket collectives."""
    sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isx.dp import GradAllReducer, broadcast_module_state
    torch.manual_seed(7 + rank)
    net = _TinyNet()
    broadcast_module_state(net)
    r = GradAllReducer(list(net.parameters()), bucket_mb=0.00001)          # one bucket per parameter tensor
    assert len(r.buckets) > 2
    net.train()
    x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(3))
    res = {}
    # step 1: only rank 0 computes
    r.zero_grad()
    if rank == 0:
        r.arm()
        net(x, x, x)[0].sum().backward()
    r.finish()
    res["step1"] = r.flat.clone()
    # step 2: somebody dropped the gradients (optimizer.zero_grad(set_to_none=True)); both ranks compute
    for p in net.parameters():
        p.grad = None
    r.arm()
    net(x, x, x)[0].sum().backward()
    r.finish()
    res["step2"] = r.flat.clone()
    res["views"] = all(r._is_view(p) for p in r.params)
    # step 3: gradients dropped again, NO reducer.zero_grad(), and the head takes no part in this backward: its slice of the flat
    # buffer still holds step 2's summed gradient -- it must be exchanged (and handed to the optimizer) as zeros, not once more
    for p in net.parameters():
        p.grad = None
    r.arm()
    net.features(x).sum().backward()
    r.finish()
    lo, hi = r.slices[net.head.weight]
    res["step3_head"] = r.flat[lo:hi].clone()
    lo, hi = r.slices[net.features[0].weight]
    res["step3_conv"] = r.flat[lo:hi].clone()
    res["views3"] = all(r._is_view(p) for p in r.params)
    res["w0"] = net.head.weight.detach().clone()
    torch.save(res, out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_all_reducer_uneven_ranks_and_detached_grads(tmp_path):
    out = str(tmp_path / "r")
    mp.spawn(_run_uneven_reducer, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(a["w0"], b["w0"])                                   # broadcast made the replicas identical
    assert torch.equal(a["step1"], b["step1"]) and float(a["step1"].abs().sum()) > 0
    assert torch.equal(a["step2"], b["step2"]) and a["views"] and b["views"]
    # step 2 = both ranks' (identical) gradients summed = 2 x step 1's single contribution
    np.testing.assert_allclose(a["step2"].numpy(), 2 * a["step1"].numpy(), rtol=1e-6, atol=1e-7)
    assert float(a["step3_head"].abs().sum()) == 0.0 and float(b["step3_head"].abs().sum()) == 0.0      # unused parameter: no stale gradient
    assert torch.equal(a["step3_conv"], b["step3_conv"]) and float(a["step3_conv"].abs().sum()) > 0 and a["views3"] and b["views3"]


def test_region_training_runs_and_learns_on_cpu():
    """siamese_regions training (triplet + window classification loss, micro-batch 1) end to end on a tiny set."""
    from train import siamese_regions as sr
    from utils.dataset import synthetic_image_set
    torch.manual_seed(0); random.seed(0)
    P = sr.P
    P.cuda_device, P.cnn_model, P.feature_size2d, P.feature_dim, P.regions_k = -1, "alexnet", (6, 6), 16, 3
    P.train_epochs, P.train_batch_size, P.test_batch_size, P.train_loss_int, P.untrained_blocks = 1, 4, 4, 1000, 4
    P.train_lr, P.train_epoch_switch = 1e-3, 1
    tr = synthetic_image_set(8, 2, size=(3, 288, 288), seed=1)
    te = synthetic_image_set(4, 2, size=(3, 288, 288), seed=2)
    before = None
    net, score = sr.main(tr, tr, te)
    assert score >= 0 and any(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in net.parameters() if p.requires_grad)


def test_grad_all_reducer_single_process_is_noop():
    from isx.dp import GradAllReducer
    net = _TinyNet()
    r = GradAllReducer(list(net.parameters()), bucket_mb=0.0001)
    assert len(r.buckets) > 1 and r.flat.numel() == sum(p.numel() for p in net.parameters())
    r.zero_grad()
    x = torch.randn(2, 3, 8, 8)
    net.train()
    r.arm()
    net(x, x, x)[0].sum().backward()
    r.finish()
    assert all(p.grad.data_ptr() >= r.flat.data_ptr() for p in net.parameters())          # grads live in the flat buffer
    assert float(r.flat.abs().sum()) > 0


# ------------------------------------------------------------------ GPU kernels
@pytest.mark.gpu
def test_mining_and_triplet_kernels(golden):
    from isx import ops
    from model.custom_modules import TripletLoss
    g = golden("training.npz")
    dev = lambda a: t(a).cuda()
    for semi in (1, 0):
        neg = ops.mine_negatives(dev(g["mine_sim"]), dev(g["mine_labels"]), dev(g["mine_i1"]), dev(g["mine_i2"]), semi)
        np.testing.assert_array_equal(neg.cpu().numpy(), g["neg_semi%d" % semi])
    rng = np.random.default_rng(0)
    N = 3000
    E = O.l2norm_rows(rng.standard_normal((N, 64), dtype=np.float32))
    E[100] = E[7]
    sim = O.cosine_sim(E, E)
    lab = (np.arange(N) % 300).astype(np.int32)
    i1 = rng.integers(0, N, 500); i2 = (i1 + 300 * rng.integers(0, 9, 500)) % N
    for semi in (1, 0):
        neg = ops.mine_negatives(dev(sim), dev(lab), dev(i1), dev(i2), semi)
        np.testing.assert_array_equal(neg.cpu().numpy(), O.mine_negatives(sim, lab, i1, i2, semi))
    B, D = 37, 2048
    a, p, n = (O.l2norm_rows(rng.standard_normal((B, D), dtype=np.float32)) for _ in range(3))
    p[:10] = O.l2norm_rows(a[:10] + 0.01 * rng.standard_normal((10, D), dtype=np.float32))
    for normalized in (True, False):
        lo, rows, ga, gp, gn = O.triplet_loss(a, p, n, 0.1, normalized, True)
        got = ops.triplet_loss_rows(dev(a), dev(p), dev(n), 0.1, normalized)
        np.testing.assert_allclose(got.cpu().numpy(), rows, rtol=1e-5, atol=1e-6)
        ar, pr, nr = (dev(x).requires_grad_(True) for x in (a, p, n))
        loss = TripletLoss(0.1, True, normalized)(ar, pr, nr)
        assert abs(float(loss) - lo) <= 1e-5
        loss.backward()
        on = got.cpu().numpy() > 0
        np.testing.assert_allclose(ar.grad.cpu().numpy()[on], ga[on], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(pr.grad.cpu().numpy()[on], gp[on], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(nr.grad.cpu().numpy()[on], gn[on], rtol=1e-6, atol=1e-7)
        assert float(ar.grad[torch.from_numpy(~on).cuda()].abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("L,k,D", [(8, 8, 2048), (3, 5, 100), (1, 1, 64), (2, 13, 2052)])
def test_triplet_loss_of_all_micro_batches_in_one_launch(L, k, D):
    """isx_triplet_leaves against the per-micro-batch path it replaces (TripletLoss module: isx_triplet_loss_fwd + torch sum + isx_triplet_loss_bwd_dev
    per leaf, the weight of the leaf handed over by autograd): gradient rows BIT-identical, per-leaf losses equal to the oracle's rows added in row
    order, for both loss forms, averaged or not."""
    from isx import ops
    from model.custom_modules import TripletLoss
    rng = np.random.default_rng(L * 100 + k)
    d = O.l2norm_rows(rng.standard_normal((L * 3 * k, D), dtype=np.float32))
    d[k:k + max(1, k // 2)] = O.l2norm_rows(d[:max(1, k // 2)] + 0.01 * rng.standard_normal((max(1, k // 2), D), dtype=np.float32))    # easy positives: clamped rows
    dev = torch.from_numpy(d).cuda()
    share = k / float(L * k)
    for normalized in (True, False):
        for avg in (True, False):
            loss, dd = ops.triplet_leaves(dev, L, 0.1, normalized, (1.0 / k) if avg else 1.0, share)
            want_dd = torch.empty_like(dev)
            for j in range(L):
                leaf = dev[j * 3 * k:(j + 1) * 3 * k].detach().requires_grad_(True)
                a, p, n = leaf[:k], leaf[k:2 * k], leaf[2 * k:]
                obj = TripletLoss(0.1, avg, normalized)(a, p, n) * share
                obj.backward()
                want_dd[j * 3 * k:(j + 1) * 3 * k] = leaf.grad
                _, rows, _, _, _ = O.triplet_loss(d[j * 3 * k:j * 3 * k + k], d[j * 3 * k + k:j * 3 * k + 2 * k], d[j * 3 * k + 2 * k:(j + 1) * 3 * k], 0.1, normalized, avg)
                got_rows = ops.triplet_loss_rows(a.detach(), p.detach(), n.detach(), 0.1, normalized).cpu().numpy()
                acc = np.float32(0)
                for r in got_rows:
                    acc = np.float32(acc + r)
                assert float(loss[j]) == float(acc)                              # the kernel's own rows, added in row order
                np.testing.assert_allclose(float(loss[j]), rows.sum(), rtol=1e-5, atol=1e-6)
            assert torch.equal(dd, want_dd), (normalized, avg)


@pytest.mark.gpu
def test_training_epoch_on_gpu_reduces_loss():
    from train import siamese_descriptor as sd
    from utils.dataset import synthetic_image_set
    torch.manual_seed(0); random.seed(0)
    P = sd.P
    P.cuda_device, P.cnn_model, P.feature_size2d, P.feature_dim = 0, "alexnet", (6, 6), 32
    P.train_epochs, P.train_batch_size, P.train_micro_batch, P.test_batch_size = 2, 8, 4, 16
    P.train_loss_int, P.untrained_blocks, P.train_epoch_switch, P.train_lr = 1000, 4, 1, 1e-3
    tr = synthetic_image_set(24, 4, seed=1)
    te = synthetic_image_set(8, 4, seed=2)
    net, score = sd.main(tr, tr, te)
    assert next(net.parameters()).is_cuda and score >= 0


def test_siamese_branches_share_one_trunk_pass():
    """forward(x1, x2, x3) in training mode == three forward_single calls (eval-mode BN: samples are independent)."""
    from isx import backbones
    from model.siamese import DescriptorNet, TuneClassif
    torch.manual_seed(0)
    net = DescriptorNet(TuneClassif(backbones.alexnet(pretrained=True, seed=0), 5), 16, (6, 6), untrained=-1)
    net.train()
    xs = [torch.randn(2, 3, 224, 224) for _ in range(3)]
    a, p, n = net(*xs)
    for got, x in zip((a, p, n), xs):
        torch.testing.assert_close(got, net.forward_single(x), rtol=1e-5, atol=1e-6)
    a2, p2 = net(xs[0], xs[1])
    torch.testing.assert_close(a2, a, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_resident_images_and_frozen_trunk_training_step():
    """Device-resident datasets gather exactly the staged batch; behind a frozen trunk the head's Shift parameter and
    Linear both receive gradients (the fused inference head must not be used while they train)."""
    from isx import backbones
    from model.custom_modules import TripletLoss
    from model.siamese import DescriptorNet, TuneClassif
    from train import _common as TC
    g = torch.Generator().manual_seed(4)
    ds = [(torch.randn(3, 224, 224, generator=g), "l%d" % (i % 3), "p%d" % i) for i in range(12)]
    raw = [(torch.randint(0, 256, (40, 30, 3), generator=g, dtype=torch.uint8), "a", "r%d" % i) for i in range(6)]
    TC.drop_resident()
    try:
        batch = [ds[i] for i in (5, 0, 7, 7, 2)]
        want = TC.stage_batch(batch, None, 0)                       # stack + H2D
        assert TC.make_resident(ds, 0) is not None
        got = TC.stage_batch(batch, None, 0)                        # device-side row gather
        assert torch.equal(got, want)
        TC.RAW_INGEST["mean"], TC.RAW_INGEST["std"] = [0.4, 0.5, 0.6], [0.2, 0.25, 0.3]
        want_raw = TC.stage_batch(raw[:4], None, 0)
        TC.make_resident(raw, 0)
        assert torch.equal(TC.stage_batch(raw[:4], None, 0), want_raw)
    finally:
        TC.drop_resident()
        TC.RAW_INGEST["mean"] = TC.RAW_INGEST["std"] = None
    torch.manual_seed(0)
    net = DescriptorNet(TuneClassif(backbones.resnet18(pretrained=True, seed=0), 5), 32, (7, 7), untrained=-1).cuda()
    net.train()
    for m in net.features.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    xs = [torch.randn(4, 3, 224, 224, device="cuda") for _ in range(3)]
    a, p, n = net(*xs)
    TripletLoss(0.1, False, True)(a, p, n).backward()
    shift, lin = net.feature_reduc1[1], net.feature_reduc1[2]
    assert shift.param.grad is not None and float(shift.param.grad.abs().sum()) > 0
    assert lin.weight.grad is not None and float(lin.weight.grad.abs().sum()) > 0
    # the frozen trunk went through the folded inference kernels: same descriptors as the plain trunk within fp32 noise
    net._trunk.folded = None
    ref = net.forward_single(xs[0])
    plain = net.feature_reduc2(net.feature_reduc1(net.features(xs[0]).reshape(4, -1)))
    torch.testing.assert_close(ref, plain, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_minibatch_trunk_precompute_is_bit_identical():
    """Frozen-trunk siamese training (round 3): the trunk of a whole mini-batch runs as ONE launch and the micro-batches run the head on their
    rows of the features -- the same weights, bit for bit, as one trunk launch per micro-batch (P.train_trunk_per_minibatch = False), after two
    epochs of SGD with gradient accumulation, semi-hard then hard mining and random fall-back negatives.  (ResNet-50: the whole trunk runs in
    libisx, whose outputs do not depend on the launch size; a trunk with MIOpen-run layers -- ResNet-18's strided 1x1 shortcuts, AlexNet --
    is only equal up to MIOpen's per-size algorithm choice.)"""
    import copy
    from train import siamese_descriptor as sd
    from utils.dataset import synthetic_image_set
    saved = copy.copy(sd.P.__dict__)
    tr = synthetic_image_set(32, 4, seed=1, structure=0.5)
    te = synthetic_image_set(8, 4, seed=2, structure=0.5)
    out = {}
    try:
        for per_minibatch in (True, False):
            torch.manual_seed(0); random.seed(0)
            P = sd.P
            P.cuda_device, P.cnn_model, P.feature_size2d, P.feature_dim = 0, "resnet50", (7, 7), 32      # every convolution in libisx: batch-invariant bits
            P.train_epochs, P.train_batch_size, P.train_micro_batch, P.test_batch_size = 2, 12, 4, 16
            P.train_loss_int, P.untrained_blocks, P.train_epoch_switch, P.train_lr, P.train_pre_proc = 1000, -1, 1, 1e-2, True
            P.train_trunk_per_minibatch = per_minibatch
            net, _ = sd.main(tr, tr, te)
            assert net.trunk_precomputable() or not net.training
            out[per_minibatch] = {k: v.detach().clone() for k, v in net.state_dict().items()}
    finally:
        sd.P.__dict__.clear(); sd.P.__dict__.update(saved)
    a, b = out[True], out[False]
    assert set(a) == set(b)
    moved = 0.0
    for k in a:
        assert torch.equal(a[k], b[k]), k
    torch.manual_seed(0)
    assert any("feature_reduc1" in k for k in a)


def test_untrained_blocks_follow_the_reference_table():
    """train/*_p.py:14-17,48 of the reference: untrained_blocks[cnn_model.lower()] -- layer4 of a ResNet (conv5 of AlexNet) is TRAINED."""
    from isx import backbones
    from model.siamese import DescriptorNet, TuneClassif, first_trainable
    from train.params import Params, UNTRAINED_BLOCKS
    assert UNTRAINED_BLOCKS["alexnet"] == 4 and UNTRAINED_BLOCKS["resnet152"] == 2 + 3 + 8 + 36 and UNTRAINED_BLOCKS["resnet50"] == 15
    P = Params()
    assert P.untrained_blocks == 4                        # AlexNet default
    P.cnn_model = "ResNet152"
    assert P.untrained_blocks == 49
    P.untrained_blocks = -1
    assert P.untrained_blocks == -1                       # an assigned value wins
    P.untrained_blocks = None
    P.cnn_model = "resnet50"
    net = DescriptorNet(TuneClassif(backbones.resnet50(pretrained=True, seed=0), 5, untrained=P.untrained_blocks), 16, (7, 7),
                        untrained=P.untrained_blocks)
    names = [n for n, p in net.named_parameters() if p.requires_grad and n.startswith("features.")]
    split = first_trainable(net.features)
    assert split == 4 + 3 + 4 + 6                          # conv1, bn1, relu, maxpool, then layers 1-3: the first block of layer4
    assert names and all(int(n.split(".")[1]) >= split for n in names)
    assert sum(1 for _ in net.features[split:]) == 3       # layer4 = 3 bottlenecks
    frozen = DescriptorNet(TuneClassif(backbones.resnet50(pretrained=True, seed=0), 5), 16, (7, 7), untrained=-1)
    assert first_trainable(frozen.features) == len(frozen.features)


def _train_reference_config(split_trunk, suffix_engine, epochs, n_images, batch, micro, mined, replay, batched=True, feature_dim=32, prefix_cache=False,
                            prefix_ahead=None, calls=None):
    """One run of train.siamese_descriptor.main on the reference configuration (ResNet-50, untrained_blocks from the table).  `mined`:
    list receiving the mined negatives per epoch; `replay`: a previous run's list to use instead of mining."""
    import copy
    import model.siamese as ms
    from train import siamese_descriptor as sd
    from utils.dataset import synthetic_image_set
    saved = copy.copy(sd.P.__dict__)
    tr = synthetic_image_set(n_images, 4, seed=1, structure=0.5)
    te = synthetic_image_set(8, 4, seed=2, structure=0.5)
    real_mine = sd.mine_epoch_negatives
    old = (ms.SPLIT_TRUNK, ms.SUFFIX_ENGINE)
    try:
        ms.SPLIT_TRUNK, ms.SUFFIX_ENGINE = split_trunk, suffix_engine
        if replay is None:
            sd.mine_epoch_negatives = lambda *a, **k: (mined.append(real_mine(*a, **k)) or mined[-1])
        else:
            it = iter(replay)
            sd.mine_epoch_negatives = lambda *a, **k: next(it)
        P = sd.P
        P.cuda_device, P.cnn_model, P.feature_size2d, P.feature_dim = 0, "resnet50", (7, 7), feature_dim
        P.train_epochs, P.train_batch_size, P.train_micro_batch, P.test_batch_size = epochs, batch, micro, 16
        P.train_loss_int, P.train_epoch_switch, P.train_lr, P.train_pre_proc = 1000, 1, 1e-3, True
        P.untrained_blocks = None                                  # the reference's table: 15 for ResNet-50
        P.train_suffix_batched = batched
        P.train_prefix_cache = prefix_cache
        if prefix_ahead is not None:
            P.train_prefix_ahead = prefix_ahead
        if calls is not None:                                      # how many images each prefix launch of the run carried
            real_prefix = ms._SplitTrunk.prefix
            ms._SplitTrunk.prefix = lambda self_, features, x: (calls.append(int(x.size(0))) or real_prefix(self_, features, x))
        assert P.untrained_blocks == 15
        torch.manual_seed(0); random.seed(0)
        init = {k: v.detach().clone() for k, v in sd.get_siamese_net().state_dict().items()}
        torch.manual_seed(0); random.seed(0)
        net, _ = sd.main(tr, tr, te)
        assert net.trunk_precomputable() == split_trunk
        if prefix_cache:
            st = net._trunk.cache_stats
            assert st["rows_computed"] <= n_images and st["rows_served"] > 4 * st["rows_computed"], st      # every image computed once, served many times
        return init, {k: v.detach().clone() for k, v in net.state_dict().items()}
    finally:
        ms.SPLIT_TRUNK, ms.SUFFIX_ENGINE = old
        sd.mine_epoch_negatives = real_mine
        if calls is not None:
            ms._SplitTrunk.prefix = real_prefix
        sd.P.__dict__.clear(); sd.P.__dict__.update(saved)


def _weight_deviation(a, b, init):
    """(max over the WEIGHT tensors (convolution / linear weights: >= 2-d) of max|a - b| / max|b|,
        max over the tensors that moved of max|a - b| / max|b - init|  [the deviation relative to the UPDATE],
        names of the two worst tensors, how far layer4 and the head moved); asserts that the frozen prefix did not move."""
    worst_w, worst_u, name_w, name_u, moved4, moved_head = 0.0, 0.0, None, None, 0.0, 0.0
    for k in a:
        if not a[k].dtype.is_floating_point:
            continue
        d = float((b[k] - init[k].to(b[k].device)).abs().max())
        err = float((a[k] - b[k]).abs().max())
        if a[k].dim() >= 2 and float(init[k].abs().max()) > 0.0 and err / float(b[k].abs().max()) > worst_w:
            worst_w, name_w = err / float(b[k].abs().max()), k
        if d > 0 and err / d > worst_u:
            worst_u, name_u = err / d, k
        if k.split(".")[0] == "features" and int(k.split(".")[1]) >= 17:
            moved4 = max(moved4, d)
        elif k.startswith("features."):
            assert d == 0.0, k                                     # frozen prefix untouched
        else:
            moved_head = max(moved_head, d)
    return worst_w, worst_u, name_w, name_u, moved4, moved_head


@pytest.mark.gpu
def test_reference_config_one_step_matches_plain_torch_training():
    """The reference's training configuration (stem + layers 1-3 frozen, layer4 + head trained): frozen prefix on the BN-folded HIP trunk
    without a graph, layer4 forward AND backward on the libisx suffix engine, head weight gradient from the step's rows -- against the PLAIN
    torch run (whole trunk = features(x) under autograd, MIOpen).  After ONE optimizer step (5 micro-batches accumulated, SGD with momentum
    and weight decay) every convolution / linear weight agrees to <= 1e-4 of its tensor's scale (measured 5e-6 ... 1.6e-5); the 1-d parameters (biases, BatchNorm affine,
    Shift offsets: they start at or near zero) are judged against the size of the update, see below."""
    mined = []
    init, a = _train_reference_config(True, True, 1, 16, 40, 8, mined, None)     # 16 images, 4 labels: 40 positive couples = ONE mini-batch of 5 micro-batches
    _, b = _train_reference_config(False, False, 1, 16, 40, 8, None, mined)
    worst_w, worst_u, name_w, name_u, moved4, moved_head = _weight_deviation(a, b, init)
    print("reference config, one step, HIP prefix + suffix engine vs plain torch: max |dw| / max|w| = %.3g (%s); relative to the update %.3g (%s); "
          "layer4 moved %.3g, head moved %.3g" % (worst_w, name_w, worst_u, name_u, moved4, moved_head))
    assert moved4 > 0 and moved_head > 0
    # 1e-6 of the weight scale would need identical ReLU patterns in both runs -- and a reproducible plain run: MIOpen's split-K weight-gradient
    # kernels (igemm_wrw ... gkgs: atomic adds) move this number between 5e-6 and 1.6e-5 from one run of the SAME plain configuration to the next
    assert worst_w <= 1e-4
    # relative to the UPDATE itself the runs differ by per cents in the worst tensor -- always a BatchNorm bias or a Shift offset: a column sum
    # over the rows whose ReLU is open, and a pre-activation of size 1e-7 lands on either side of 0 depending on the rounding of the forward pass
    # (folded vs unfolded BatchNorm, k-ordered chain vs MIOpen's blocked sums); tests/test_gpu_suffix.py pins the masks and finds 2e-6
    assert worst_u <= 0.25


@pytest.mark.gpu
def test_reference_config_two_epochs_track_plain_torch_training():
    """The same comparison after two epochs (36 steps, semi-hard then hard mining; the second run replays the first run's mined negatives so
    that the comparison is about arithmetic, not about an arg-max flipping).  Training is a chaotic map: a pre-activation that rounds to the
    other side of a ReLU, or a loss term that crosses the margin, sends the runs apart at a rate no forward tolerance controls -- torch's own
    fp32 path against itself with a different MIOpen algorithm does the same.  The test states what holds: the suffix engine and the plain
    suffix behind the same HIP prefix stay within half of the update of every tensor, both move layer4 and the head, the prefix stays frozen."""
    mined = []
    init, a = _train_reference_config(True, True, 2, 32, 12, 4, mined, None)
    _, b = _train_reference_config(True, False, 2, 32, 12, 4, None, mined)
    _, c = _train_reference_config(False, False, 2, 32, 12, 4, None, mined)
    worst, worst_u, name, name_u, moved4, moved_head = _weight_deviation(a, b, init)
    worst_plain, worst_plain_u, name_plain, _, _, _ = _weight_deviation(a, c, init)
    print("reference config, two epochs: suffix engine vs torch suffix (same HIP prefix): %.3g of the weight scale (%s), %.3g of the update (%s); vs the "
          "plain torch run %.3g / %.3g (%s); layer4 moved %.3g, head moved %.3g"
          % (worst, name, worst_u, name_u, worst_plain, worst_plain_u, name_plain, moved4, moved_head))
    assert moved4 > 0 and moved_head > 0
    # a tracking check, not a parity check: measured 5e-4 ... 7e-2 of the weight scale depending on which ReLU units / margin crossings flip
    # (the head's weights are mostly update after 36 steps: they move 100 x their initial scale)
    assert worst_u <= 0.5

@pytest.mark.gpu
def test_batched_suffix_is_bit_identical_to_leaf_by_leaf():
    """utils/train_general._Stepper._leaves_batched: all micro-batches of a step through the suffix engine AND the head engine in ONE forward /
    backward each, their gradients kept apart, against the same engines driven micro-batch by micro-batch (what a rank of an 8-GPU run does
    with its single leaf).
    Two epochs of SGD on the reference configuration (NO replay of mined negatives: the runs must not differ at all): every tensor of the
    state dict bit-identical -- the property that makes the update independent of the number of ranks."""
    mined = []
    init, a = _train_reference_config(True, True, 2, 32, 12, 4, mined, None, batched=True)
    _, b = _train_reference_config(True, True, 2, 32, 12, 4, [], None, batched="leaf")
    moved = 0.0
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k].float() - b[k].float()).abs().max()))
        if a[k].dtype.is_floating_point:
            moved = max(moved, float((a[k] - init[k].to(a[k].device)).abs().max()))
    assert moved > 0


def _train_reference_two_ranks_one_gpu(rank, world, port, out, batch=16, micro=4, feature_dim=32):
    """worker of test_ranks_on_one_gpu_match_single_process: rank `rank` of `world`, every rank on cuda:0, gloo process group"""
    sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, state = _train_reference_config(True, True, 2, 32, batch, micro, [], None, feature_dim=feature_dim)
    if rank == 0:
        from isx import dp as _dp
        torch.save({k: v.cpu() for k, v in state.items()}, out)
        torch.save(dict(_dp.STATS), out + ".stats")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_prefix_feature_cache_trains_the_same_bits():
    """P.train_prefix_cache: the frozen prefix's features of the resident training images come from an HBM table (each image computed once, in
    launches of its own) instead of being recomputed at every use (reference utils/train_general.py:51-74: the trunk runs per micro-batch) --
    two epochs on the reference configuration end in the SAME state dict, bit for bit: the kernels are batch-invariant."""
    _, a = _train_reference_config(True, True, 2, 32, 16, 4, [], None)
    _, b = _train_reference_config(True, True, 2, 32, 16, 4, [], None, prefix_cache=True)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
def test_prefix_look_ahead_trains_the_same_bits():
    """P.train_prefix_ahead: the frozen prefix of three consecutive mini-batches in ONE launch (the features of an image depend neither on its
    batch nor on the optimizer steps in between) against every step launching its own -- the same state dict after two epochs, bit for bit; the
    launches did carry 3 x 48 images (a trailing block of two carries 96, a single step launches its own 48)."""
    from utils.dataset import get_pos_couples, synthetic_image_set
    steps = sum(len(v) for v in get_pos_couples(synthetic_image_set(32, 4, seed=1, structure=0.5)).values()) // 16
    assert steps >= 4
    own, ahead = [], []
    _, a = _train_reference_config(True, True, 2, 32, 16, 4, [], None, prefix_ahead=1, calls=own)
    _, b = _train_reference_config(True, True, 2, 32, 16, 4, [], None, prefix_ahead=3, calls=ahead)
    assert own.count(48) == 2 * steps and 144 not in own
    rest = steps % 3
    assert ahead.count(144) == 2 * (steps // 3) and ahead.count(96) == (2 if rest == 2 else 0) and ahead.count(48) == (2 if rest == 1 else 0)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("world,batch,micro,feature_dim", [(2, 16, 4, 32), (4, 64, 8, 32), (4, 64, 8, 256)])
def test_ranks_on_one_gpu_match_single_process(tmp_path, world, batch, micro, feature_dim):
    """Data-parallel training of the REAL net with the REAL kernels: DescriptorNet(ResNet-50) on the reference configuration, `world` ranks
    (all on cuda:0, gloo -- the box has one GPU; RCCL replaces only the transport) x their subtree of the micro-batches of every step, HIP
    prefix + batched suffix engine + head engine + row-deferred head gradient + TreeExchange, against ONE process with all micro-batches: the
    whole state dict BIT-IDENTICAL after two epochs (mining included: no replay).  (4, 64, 8) is the reference's step -- 64 triplets as 8
    micro-batches of 8: one process runs 192 rows per pass (192-row tiles in the head GEMMs), each of the 4 ranks 48 (64-row tiles).
    feature_dim 256: a head wide enough for the 8 feature groups -- the ranks run it SHARDED by output features (isx/shard_head.py: each rank its
    64 features' slice of the forward for all rows, its rows of the fused gradient + SGD kernel, its groups' chains of the input gradient), the
    single process the plain kernels: the same bits."""
    out = str(tmp_path / "dp.pt")
    mp.spawn(_train_reference_two_ranks_one_gpu, args=(world, _free_port(), out, batch, micro, feature_dim), nprocs=world, join=True)
    b = torch.load(out)
    st = torch.load(out + ".stats")
    assert (st.get("head_shard_bytes_received", 0) > 0) == (feature_dim % 256 == 0)
    init, a = _train_reference_config(True, True, 2, 32, batch, micro, [], None, feature_dim=feature_dim)
    moved = 0.0
    for k in a:
        assert torch.equal(a[k].cpu(), b[k]), (k, float((a[k].cpu().float() - b[k].float()).abs().max()))
        if a[k].dtype.is_floating_point:
            moved = max(moved, float((a[k] - init[k].to(a[k].device)).abs().max()))
    assert moved > 0


@pytest.mark.gpu
@pytest.mark.parametrize("model,fs,size", [("alexnet", (6, 6), 288), ("resnet50", (7, 7), 288)])
def test_region_training_runs_on_gpu(model, fs, size):
    """siamese_regions training (reference train/siamese_regions.py: triplet + window classification loss, micro-batch 1, the head shared by
    the k windows of an image) on the GPU: the region kernels on the no-graph side, autograd on the training side, the row-deferred head
    gradient collecting the rows of every window -- runs, and the trainable parameters receive gradients."""
    import copy
    from train import siamese_regions as sr
    from utils.dataset import synthetic_image_set
    saved = copy.copy(sr.P.__dict__)
    try:
        torch.manual_seed(0); random.seed(0)
        P = sr.P
        P.cuda_device, P.cnn_model, P.feature_size2d, P.feature_dim, P.regions_k = 0, model, fs, 16, 3
        P.train_epochs, P.train_batch_size, P.test_batch_size, P.train_loss_int = 1, 4, 4, 1000
        P.untrained_blocks = None                                  # the reference's table
        # (a seeded random-init ResNet-50 with identity BatchNorm statistics has activations of size 1e3 at layer4: its class-score loss needs a
        # tiny step to stay finite -- the test is about the plumbing, not about learning on noise)
        P.train_lr, P.train_epoch_switch = (1e-3 if model == "alexnet" else 1e-9), 1
        tr = synthetic_image_set(8, 2, size=(3, size, size), seed=1)
        te = synthetic_image_set(4, 2, size=(3, size, size), seed=2)
        net, score = sr.main(tr, tr, te)
        assert next(net.parameters()).is_cuda and score >= 0
        lin = net.feature_reduc1[2]
        assert lin.weight.grad is not None and bool(torch.isfinite(lin.weight.grad).all()) and float(lin.weight.grad.abs().sum()) > 0     # from the windows' (x, dy) rows
        assert any(p.grad is not None and float(p.grad.abs().sum()) > 0 for n, p in net.named_parameters() if n.startswith("features.") and p.requires_grad)
    finally:
        sr.P.__dict__.clear(); sr.P.__dict__.update(saved)


def test_training_entry_point_reads_its_dataset_like_the_reference(monkeypatch, capsys):
    """`run()` / `python -m train.siamese_descriptor`: the sets come from P.dataset_full (here a `synthetic:` spec), the dataset-dependent fields of P
    are filled from the dataset id, P.test_upfront / P.train are honoured, and the command-line form sets the same fields."""
    import copy
    from train import _common as TC
    from train import siamese_descriptor as sd
    saved = copy.copy(sd.P.__dict__)
    calls = []
    monkeypatch.setattr(sd, "train_siam_triplets_pos_couples", lambda net, train_set, testset_tuple, *a, **k: calls.append((len(train_set), len(testset_tuple[0]))) or 7)
    monkeypatch.setattr(sd, "test_print_descriptor", lambda *a, **k: calls.append("test") or 3)
    monkeypatch.setattr(sd, "get_siamese_net", lambda: nn.Linear(2, 2))
    try:
        P = sd.P
        P.cuda_device, P.cnn_model = -1, "alexnet"
        net, score = sd.run("synthetic:CLICIDE_video_224sq:n=12:q=4:labels=3:size=32")
        assert calls == ["test", (12, 4), "test"] and score == 7
        assert P.num_classes == 3 and P.feature_size2d == (6, 6) and tuple(P.image_input_size) == (3, 224, 224) and sd.labels == sorted(sd.labels) and len(sd.labels) == 3
        del calls[:]
        P.test_upfront, P.train = False, False
        net, score = sd.run("synthetic:CLICIDE_video_224sq:n=12:q=4:labels=3:size=32")
        assert calls == [] and score == 0
        P.test_upfront, P.train = True, True
        TC.training_cli(["--dataset=synthetic:CLICIDE_video_224sq:n=8:q=4:labels=2:size=32", "--model=alexnet", "--device=-1", "--epochs=3", "--lr=0.5"], P, sd.run,
                        "train.siamese_descriptor")
        assert calls == ["test", (8, 4), "test"] and P.train_epochs == 3 and P.train_lr == 0.5 and P.dataset_full.startswith("synthetic:")
        P.train_pre_proc = False
        with pytest.raises(NotImplementedError):
            sd.run()
    finally:
        sd.P.__dict__.clear(); sd.P.__dict__.update(saved)


def test_training_entry_point_under_a_two_rank_launch(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 <script calling train._common.training_cli>` (gloo, CPU): the command-line entry opens the
    process group, train_gen splits the micro-batches over the two ranks, and both ranks end on the same weights as ONE process run from the same
    command."""
    import subprocess
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "instance-search_amd"), OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--dataset=synthetic:CLICIDE_video_224sq:n=6:q=2:labels=2:size=224", "--model=alexnet", "--device=-1", "--epochs=1", "--batch-size=4", "--micro-batch=2",
            "--feature-dim=16", "--seed=3"]
    drv = ("import os, sys, torch\n"
           "from train import siamese_descriptor as sd\nfrom train import _common as TC\n"
           "torch.manual_seed(0); torch.set_num_threads(1)\n"
           "sd.P.test_upfront = False; sd.P.train_loss_int = 1000\n"
           "net, _ = TC.training_cli(%r, sd.P, sd.run, 'train.siamese_descriptor')\n"
           "torch.save({k: v.clone() for k, v in net.state_dict().items()}, %r + os.environ.get('RANK', 'single'))\n") % (args, str(tmp_path / "w."))
    script = tmp_path / "drv.py"
    script.write_text(drv)
    one = subprocess.run([sys.executable, str(script)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                          str(_free_port()), str(script)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    a, b0, b1 = (torch.load(str(tmp_path / ("w." + r))) for r in ("single", "0", "1"))
    for k in a:
        assert torch.equal(b0[k], b1[k]), k
        assert torch.equal(a[k], b0[k]), k
    assert any("feature_reduc1" in k for k in a)
