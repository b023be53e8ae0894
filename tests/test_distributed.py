"""world_size-2 gloo (CPU) run of the sharded-gallery driver: sharded == unsharded, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "instance-search_amd"))
    from isx import retrieval as R
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    N, D, k, Mloc = 203, 24, 10, 3
    G = torch.nn.functional.normalize(torch.randn(N, D, generator=g), dim=1)
    G[150] = G[7]                                           # tie across the shard boundary
    Qall = torch.nn.functional.normalize(torch.randn(world * Mloc, D, generator=g), dim=1)
    gal = R.ShardedGallery.from_full(G)
    lo, hi = R.shard_bounds(N, world, rank)
    assert gal.shard.size(0) == hi - lo and gal.idx_base == lo
    Q = R.gather_queries(Qall[rank * Mloc:(rank + 1) * Mloc])
    assert torch.equal(Q, Qall)
    s, i = gal.search(Q, k)
    # k larger than a shard row count exercises the (-inf,-1) padding
    tiny = R.ShardedGallery(G[lo:lo + 2], lo)
    s2, i2 = tiny.search(Q, 5)
    # replicated gallery, queries split across ranks (7 queries over 2 ranks: uneven blocks)
    rs, ri = R.ReplicatedGallery(G).search(torch.cat([Qall, Qall[:1]]), k)
    torch.save({"s": s, "i": i, "s2": s2, "i2": i2, "G": G, "Q": Qall, "lo": lo, "rs": rs, "ri": ri}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_matches_unsharded(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % k)) for k in range(world)]
    assert torch.equal(r[0]["i"], r[1]["i"]) and torch.equal(r[0]["s"], r[1]["s"])      # every rank holds the merged result
    G, Q = r[0]["G"], r[0]["Q"]
    sim = Q @ G.t()
    want = sim.sort(dim=1, descending=True, stable=True)
    assert torch.equal(r[0]["i"], want.indices[:, :10])
    assert torch.equal(r[0]["s"], want.values[:, :10])
    # the canonical order also agrees with the oracle's ranking of the same score matrix
    np.testing.assert_array_equal(r[0]["i"].numpy(), O.rank_full(sim.numpy())[:, :10])
    # replicated-gallery mode: the same lists (7 queries: 6 + the first one again), on every rank
    Q7 = torch.cat([Q, Q[:1]])
    want7 = (Q7 @ G.t()).sort(dim=1, descending=True, stable=True)
    for k_ in range(world):
        assert torch.equal(r[k_]["ri"], want7.indices[:, :10]) and torch.equal(r[k_]["rs"], want7.values[:, :10])
    # tiny shards: 2 rows each -> 4 real candidates + padding
    rows = torch.cat([G[r[0]["lo"]:r[0]["lo"] + 2], G[r[1]["lo"]:r[1]["lo"] + 2]])
    ids = torch.tensor([r[0]["lo"], r[0]["lo"] + 1, r[1]["lo"], r[1]["lo"] + 1])
    sub = (Q @ rows.t())
    o = sub.sort(dim=1, descending=True, stable=True)
    assert torch.equal(r[0]["i2"][:, :4], ids[o.indices])
    assert (r[0]["i2"][:, 4:] == -1).all() and torch.isinf(r[0]["s2"][:, 4:]).all()


def test_shard_bounds_cover_rows():
    from isx.retrieval import shard_bounds
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


@pytest.mark.parametrize("main_args", [
    ["test.classif_finetune_test", "--dataset=synthetic:CLICIDE_video_224sq:n=14:q=5:labels=3:size=224:struct=60", "--model=alexnet", "--device=-1", "--classify=False", "--batch=4", "--dba=2"],
    ["test.siamese_descriptor_test", "--dataset=synthetic:CLICIDE_video_224sq:n=11:q=4:labels=3:size=224:struct=60", "--model=alexnet", "--device=-1", "--feature-dim=16", "--batch=4", "--dba=0"],
])
@pytest.mark.parametrize("sharded,world", [("0", 2), ("1", 2), ("1", 8)])
def test_evaluation_mains_under_a_multi_rank_launch_print_the_single_process_lines(tmp_path, main_args, sharded, world):
    """`python -m torch.distributed.run --nproc-per-node N -m test.<approach>_test ...` (gloo, CPU): the ranks split queries and gallery, gather the
    descriptor rows, split the metrics by query rows -- rank 0 prints what ONE process prints (same counts; mAP to the printed digits), the other
    ranks print nothing.  N = 8 (BASELINE configs[4]'s rank count): 4-5 queries and 11-14 gallery rows over eight ranks -- ranks WITHOUT a query,
    shards of one or two rows."""
    import subprocess
    pkg = os.path.join(ROOT, "instance-search_amd")
    # both runs evaluate ONE network from a weights file: layers without a file are random-initialised, and torch seeds every process differently
    sys.path.insert(0, pkg)
    from isx import backbones
    from model.siamese import DescriptorNet, TuneClassif
    torch.manual_seed(5)
    cls = TuneClassif(backbones.alexnet(pretrained=True), 3)
    net = cls if "classif_finetune" in main_args[0] else DescriptorNet(cls, 16, (6, 6))
    weights = str(tmp_path / "w.pth.tar")
    torch.save(net.state_dict(), weights)
    main_args = main_args + ["--weights=" + weights]
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = subprocess.run([sys.executable, "-m"] + main_args, env=env, cwd=pkg, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    if sharded == "1":                                           # the gallery stays sharded by rows: sharded search + sharded average precision; no DBA there
        main_args = [a if not a.startswith("--dba=") else "--dba=0" for a in main_args]
        one = subprocess.run([sys.executable, "-m"] + main_args, env=env, cwd=pkg, capture_output=True, text=True, timeout=900)
        assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
                          str(_free_port()), "-m"] + main_args + ["--sharded=" + ("True" if sharded == "1" else "False")], env=env, cwd=pkg, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    from _lines import printed_lines as pick
    assert pick(one.stdout) and pick(two.stdout) == pick(one.stdout), (one.stdout, two.stdout)
