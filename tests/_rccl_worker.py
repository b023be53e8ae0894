"""Worker of tests/test_rccl_multi_gpu.py: one process per GPU under torch.distributed.run, REAL RCCL (backend nccl).
Writes what it computed to <out>.<rank>; the pytest process compares."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402


class TinyNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.features = nn.Sequential(nn.Conv2d(3, 8, 3, stride=2), nn.BatchNorm2d(8), nn.ReLU())
        self.head = nn.Linear(8 * 7 * 7, 16)

    def forward(self, x):
        y = self.head(self.features(x).flatten(1))
        return y / (y.pow(2).sum(1, keepdim=True) + 1e-10).sqrt()


def shard_head_case(rank, world, dev, res):
    from isx import dp, ops
    # ---- the descriptor head sharded by output features (isx/shard_head.py): all-gathers of rows / column slices, all-to-all of the input
    #      gradient's group chains, per-shard fused gradient + SGD, sync of the weight -- over RCCL, against one GPU's unsharded kernels ----
    from isx import shard_head
    if dp.is_power_of_two(world) and shard_head.GROUPS % world == 0:
        from isx._lib import check, lib
        g3 = torch.Generator().manual_seed(123)
        n_out, K, Rr = 256, 1280, 6
        W0, b0 = torch.randn(n_out, K, generator=g3) * 0.01, torch.randn(n_out, generator=g3)
        x_all, dy_all = torch.randn(Rr * world, K, generator=g3).to(dev), torch.randn(Rr * world, n_out, generator=g3).to(dev)
        kw = dict(lr=1e-2, momentum=0.9, weight_decay=5e-4)
        w, bias = torch.nn.Parameter(W0.clone().to(dev)), torch.nn.Parameter(b0.clone().to(dev))
        opt = torch.optim.SGD([w], **kw)
        sh = shard_head.HeadShard(w, bias)
        rows = slice(Rr * rank, Rr * rank + Rr)
        for step in range(2):                                   # the second step runs on momentum buffers and on the synced weight
            sh.begin_step(1)
            y, ctx = sh.forward(x_all[rows], [rank])
            dx = sh.backward(ctx, dy_all[rows])
            sh.finish(opt)
            sh.sync()
        res["shard_y"], res["shard_dx"], res["shard_w"] = y.cpu(), dx.cpu(), w.detach().cpu()
        if rank == 0:
            wr = torch.nn.Parameter(W0.clone().to(dev))
            optr = torch.optim.SGD([wr], **kw)
            for step in range(2):
                y_ref = ops.head_linear(x_all, wr.detach(), b0.to(dev))
                M = x_all.size(0)
                Mp = (M + 63) // 64 * 64
                dyT = dy_all.new_zeros((n_out, Mp)); dyT[:, :M] = dy_all.t()
                dx_ref = torch.empty((Mp, K), device=dev)
                check(lib().isx_head_linear_dgrad(dyT.data_ptr(), Mp, n_out, wr.data_ptr(), K, dx_ref.data_ptr(), torch.cuda.current_stream().cuda_stream), "dgrad")
                assert dp.fused_sgd_from_rows(optr, wr, dy_all, x_all)
            res["shard_ref"] = (y_ref.cpu(), dx_ref[:M].cpu(), wr.detach().cpu())


def main():
    out = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from isx import ops, retrieval as R
    from isx.dp import GradAllReducer, broadcast_module_state
    res = {}
    if os.environ.get("ISX_WORKER_CASE") == "shard_head":                       # one-rank rehearsal of the last section on a one-GPU box
        shard_head_case(rank, world, dev, res)
        # libisx's OWN RCCL communicator, on whatever number of ranks there is (one, on a one-GPU box): librccl bound by dlopen next to the copy
        # torch uses, unique id through the torch group, ncclCommInitRank, both all-gathers of the data path launched on the caller's stream, destroy
        g0 = torch.Generator().manual_seed(5)
        rows = torch.randn(37, 2048, generator=g0).to(dev)
        s0 = torch.randn(37, 100, generator=g0).to(dev)
        i0 = torch.randint(0, 1 << 40, (37, 100), generator=g0).to(dev)
        nc = R.NativeComm()
        allr = ops.comm_allgather_rows(nc.handle, nc.nranks, rows)
        as_, ai_ = ops.shard_topk_allgather(nc.handle, nc.nranks, s0, i0)
        torch.cuda.synchronize()
        res["native_rows_ok"] = bool(torch.equal(allr[rank * 37:(rank + 1) * 37], rows) and allr.shape == (world * 37, 2048))
        res["native_topk_ok"] = bool(torch.equal(as_[rank], s0) and torch.equal(ai_[rank], i0) and as_.shape == (world, 37, 100))
        nc.close()
        torch.save(res, "%s.%d" % (out, rank))
        return dist.destroy_process_group()
    # ---- sharded gallery search: torch.distributed (RCCL) exchange and libisx's own RCCL communicator ----
    g = torch.Generator().manual_seed(0)
    N, D, k, M = 40007, 256, 100, 300
    G = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    G[N // 2 + 3] = G[5]                                     # a tie across the shard boundary
    Q = torch.nn.functional.normalize(torch.randn(M, D, generator=g), dim=1)
    Gd, Qd = ops.l2norm_rows(G.to(dev)), ops.l2norm_rows(Q.to(dev))
    lo, hi = R.shard_bounds(N, world, rank)
    for name, fast in (("fast", True), ("f32", False)):
        gal = R.ShardedGallery(Gd[lo:hi], lo, fast=fast)
        s, i = gal.search(Qd, k)
        res["dist_" + name] = (s.cpu(), i.cpu())
    assert R.exchange_backend().startswith("isx_shard_topk_allgather")        # the default exchange of an RCCL group is the C-ABI entry
    os.environ["ISX_NATIVE_COMM"] = "0"                                        # A/B: the same search over torch.distributed's all-gather
    s, i = R.ShardedGallery(Gd[lo:hi], lo).search(Qd, k)
    res["torchdist"] = (s.cpu(), i.cpu())
    del os.environ["ISX_NATIVE_COMM"]
    nc = R.NativeComm()
    assert nc.nranks == world
    gal = R.ShardedGallery(Gd[lo:hi], lo, native_comm=nc)
    s, i = gal.search(Qd, k)
    res["native"] = (s.cpu(), i.cpu())
    if rank == 0:
        us, ui = ops.cosine_topk(Qd, Gd, k)                  # unsharded, this GPU alone
        res["unsharded"] = (us.cpu(), ui.cpu())
    # full-rank average precision without gathering the gallery (isx_ap_shard_*: all-gather of the positives' keys, all-reduce of the rank histograms)
    glab = (torch.arange(N) % 97).to(torch.int32)
    qlab = (torch.arange(M) % 97).to(torch.int32)
    res["sharded_ap"] = R.ShardedGallery(Gd[lo:hi], lo).average_precisions(Qd, qlab, glab[lo:hi]).cpu()
    if rank == 0:
        res["unsharded_ap"] = ops.average_precision_sim(ops.cosine_sim(Qd, Gd), qlab.to(dev), glab.to(dev), 1).cpu()
    torch.cuda.synchronize()
    nc.close()
    # ---- the two all-gathers of the N > 1 bench step on ONE communicator: data-parallel query rows -> isx_comm_allgather_rows -> search of the own
    #      shard -> isx_shard_topk_allgather -> merge, 12 steps alternating between the main stream and a side stream with the trunk's stand-in
    #      (a large GEMM) running on the other one: every step must return the unsharded list ----
    per = M // world
    q_mine = Qd[rank * per:(rank + 1) * per].contiguous()
    assert R.exchange_backend().startswith("isx_")
    gal = R.ShardedGallery(Gd[lo:hi], lo, fast=False)
    side = torch.cuda.Stream(device=dev)
    busy = torch.randn(2048, 2048, device=dev)
    pipe = []
    for step in range(12):
        stream = side if step % 2 else torch.cuda.current_stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            q_all = R.gather_queries(q_mine)
            s, i = gal.search(q_all, k)
        other = torch.cuda.current_stream() if step % 2 else side
        with torch.cuda.stream(other):
            busy = busy @ busy * 1e-3
        torch.cuda.current_stream().wait_stream(side)
        pipe.append((s, i))
    torch.cuda.synchronize()
    assert all(torch.equal(i, pipe[0][1]) and torch.equal(s, pipe[0][0]) for s, i in pipe)
    res["one_comm"] = (pipe[-1][0].cpu(), pipe[-1][1].cpu())
    res["one_comm_rows"] = per * world
    os.environ["ISX_NATIVE_COMM"] = "0"                                        # both all-gathers through torch.distributed: again one communicator
    q_all_t = R.gather_queries(q_mine)
    del os.environ["ISX_NATIVE_COMM"]
    assert torch.equal(q_all_t, R.gather_queries(q_mine)) and torch.equal(q_all_t, Qd[:per * world])
    # replicated gallery, queries split by rank
    rs, ri = R.ReplicatedGallery(Gd).search(Qd, k)
    res["replicated"] = (rs.cpu(), ri.cpu())
    # ---- gradient exchange: P ranks x 1/P of the batch + all-reduce(SUM) == one process with the whole batch ----
    torch.manual_seed(100 + rank)                            # different initial weights per rank: the broadcast must fix that
    net = TinyNet().to(dev)
    broadcast_module_state(net)
    w0 = {n: t.detach().clone().cpu() for n, t in net.state_dict().items()}
    red = GradAllReducer(list(net.parameters()), bucket_mb=0.001)
    x = torch.randn(8 * world, 3, 16, 16, generator=torch.Generator().manual_seed(9)).to(dev)
    net.train()
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()                                         # frozen BN, as in the reference's training
    red.zero_grad()
    red.arm()
    net(x[rank * 8:(rank + 1) * 8]).sum().backward()
    red.finish()
    res["flat"] = red.flat.cpu()
    res["w0"] = w0
    if rank == 0:
        ref = TinyNet().to(dev)
        ref.load_state_dict({n: t.to(dev) for n, t in w0.items()})
        ref.train()
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()
        ref(x).sum().backward()
        res["ref_flat"] = torch.cat([p.grad.reshape(-1) for p in reversed(list(ref.parameters()))]).cpu()     # the flat buffer is in backward (reverse parameter) order
    # ---- canonical gradient tree across the ranks (isx/dp.TreeExchange: RCCL all-to-all + all-gather) and the row exchange of the head ----
    from isx import dp
    if dp.is_power_of_two(world):
        g2 = torch.Generator().manual_seed(77)
        leaves = [torch.randn(100003, generator=g2) * (10.0 ** (i % 5 - 2)) for i in range(2 * world)]      # 2 leaves per rank, a length that needs padding
        lo, hi = dp.rank_leaves(len(leaves), world, rank)
        mine = dp.tree_sum(lo, hi, lambda i: leaves[i].clone().to(dev))
        dp.TreeExchange().allreduce_(mine)
        res["tree"] = mine.cpu()
        res["tree_ref"] = dp.tree_sum(0, len(leaves), lambda i: leaves[i].clone())
        w = torch.nn.Parameter(torch.zeros(16, 40, device=dev))
        sink = dp.RowSink([w])
        xr = torch.randn(3 * world, 40, generator=g2).to(dev)
        dyr = torch.randn(3 * world, 16, generator=g2).to(dev)
        sink.add(w, xr[3 * rank:3 * rank + 3], dyr[3 * rank:3 * rank + 3])
        sink.finish()
        res["rows_dw"] = w.grad.cpu()
        res["rows_ref"] = dp.weight_gradient_from_rows(dyr, xr).cpu()
    shard_head_case(rank, world, dev, res)
    torch.save(res, "%s.%d" % (out, rank))
    dist.barrier()
    torch.cuda.synchronize()
    R.close_native_comms()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
