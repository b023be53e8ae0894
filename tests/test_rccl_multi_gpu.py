"""REAL multi-GPU runs (RCCL over xGMI, one process per GPU): skipped unless the box has >= 2 GPUs.  The world_size-2
gloo tests (tests/test_distributed.py, tests/test_training.py) cover the same drivers on the CPU."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _n_gpus():
    try:
        return torch.cuda.device_count()
    except Exception:
        return 0


needs2 = pytest.mark.skipif(_n_gpus() < 2, reason="needs >= 2 GPUs (RCCL)")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ISX_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    return env


@needs2
def test_rccl_sharded_search_and_grad_allreduce(tmp_path):
    world = min(_n_gpus(), 8)
    out = str(tmp_path / "r")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_rccl_worker.py"), out]
    p = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    r = [torch.load("%s.%d" % (out, k)) for k in range(world)]
    us, ui = r[0]["unsharded"]
    for k in range(world):
        for name in ("dist_fast", "dist_f32", "torchdist", "native", "replicated"):
            s, i = r[k][name]
            assert torch.equal(i, ui), (k, name)                               # ranked lists: bit-exact, every rank, every exchange path
            assert torch.equal(s.view(torch.int32), us.view(torch.int32)), (k, name)
        s, i = r[k]["one_comm"]                                                  # query gather + result gather on ONE communicator, two streams in turn
        nq = r[k]["one_comm_rows"]
        assert torch.equal(i, ui[:nq]) and torch.equal(s.view(torch.int32), us[:nq].view(torch.int32)), k
        for n in r[0]["w0"]:
            assert torch.equal(r[k]["w0"][n], r[0]["w0"][n])                   # broadcast: identical replicas
        assert torch.equal(r[k]["flat"], r[0]["flat"])                         # every rank holds the same summed gradient
    np.testing.assert_allclose(r[0]["flat"].numpy(), r[0]["ref_flat"].numpy(), rtol=2e-5, atol=2e-6)
    for k in range(world):                                                      # sharded average precision == the unsharded kernel, float64 bits, every rank
        assert torch.equal(torch.nan_to_num(r[k]["sharded_ap"], nan=-7.0), torch.nan_to_num(r[0]["unsharded_ap"], nan=-7.0)), k
    if "tree" in r[0]:
        for k in range(world):
            assert torch.equal(r[k]["tree"], r[0]["tree_ref"]), k                # TreeExchange over RCCL == the single-process tree sum, bit for bit
    if "shard_ref" in r[0]:
        y_ref, dx_ref, w_ref = r[0]["shard_ref"]                                # the head sharded by output features == one GPU's unsharded kernels
        for k in range(world):
            n = r[k]["shard_y"].size(0)
            assert torch.equal(r[k]["shard_y"], y_ref[k * n:(k + 1) * n]) and torch.equal(r[k]["shard_dx"], dx_ref[k * n:(k + 1) * n]), k
            assert torch.equal(r[k]["shard_w"], w_ref), k
            assert torch.equal(r[k]["rows_dw"], r[k]["rows_ref"]), k             # head weight gradient from all-gathered rows == the one-process GEMM
    assert float(r[0]["flat"].abs().sum()) > 0


@needs2
def test_bench_all_gpus_bare_command_line():
    """`python bench.py --gpus N` (N = every GPU of the box, at most 8) with nothing around it: self-launch, RCCL, one JSON line."""
    n = min(_n_gpus(), 8)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "64",
           "--gallery", "4000", "--no-shard-bench"]
    p = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and lines[0].startswith('{"bench_detail"') and len(lines[1]) <= 6144
    d = json.loads(lines[1])
    assert "isx_shard_topk_allgather" in d["exchange"]["implementation"] and d["exchange"]["overlap_identical"] is True
    assert "isx_comm_allgather_rows" in d["exchange"]["implementation"] and d["exchange"]["communicators_in_data_path"] == 1
    assert d["n_gpus"] == n and d["config"]["collective_backend"] == "nccl" and d["config"]["ranks"] == n and d["value"] > 0


def test_shard_head_worker_one_rank(tmp_path):
    """The sharded-head section of the worker on ONE rank (an RCCL group of one): keeps that code exercised on a one-GPU box -- and opens libisx's own
    RCCL communicator there (one rank) and runs both data-path all-gathers on it: the library binding the N > 1 bench relies on, short of a second GPU."""
    if _n_gpus() < 1:
        pytest.skip("no GPU")
    out = str(tmp_path / "r")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_rccl_worker.py"), out]
    p = subprocess.run(cmd, env=dict(_env(), ISX_WORKER_CASE="shard_head"), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = torch.load(out + ".0")
    y_ref, dx_ref, w_ref = r["shard_ref"]
    assert torch.equal(r["shard_y"], y_ref) and torch.equal(r["shard_dx"], dx_ref) and torch.equal(r["shard_w"], w_ref)
    # libisx's own RCCL communicator (dlopen of librccl, ncclCommInitRank, isx_comm_allgather_rows, isx_shard_topk_allgather) works in this process
    assert r["native_rows_ok"] and r["native_topk_ok"]


@needs2
@pytest.mark.parametrize("sharded", ["0", "1"])
def test_evaluation_main_over_rccl_prints_the_single_process_lines(tmp_path, sharded):
    """`torch.distributed.run --nproc-per-node N -m test.classif_finetune_test` with one rank per GPU over RCCL (class scores as descriptors): the
    lines of one process, from the same weights file."""
    sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
    from isx import backbones
    from model.siamese import TuneClassif
    world = min(_n_gpus(), 8)
    torch.manual_seed(3)
    weights = str(tmp_path / "w.pth.tar")
    torch.save(TuneClassif(backbones.resnet50(pretrained=True, seed=0), 5).state_dict(), weights)
    args = ["test.classif_finetune_test", "--dataset=synthetic:CLICIDE_video_224sq:n=70:q=21:labels=5:size=224:struct=50", "--model=resnet50", "--device=0",
            "--classify=True", "--batch=16", "--dba=" + ("0" if sharded == "1" else "3"), "--weights=" + weights]
    env = dict(_env(), OMP_NUM_THREADS="1", ISX_EVAL_SHARDED=sharded)
    pkg = os.path.join(ROOT, "instance-search_amd")
    one = subprocess.run([sys.executable, "-m"] + args, env=env, cwd=pkg, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
                           str(_free_port()), "-m"] + args, env=env, cwd=pkg, capture_output=True, text=True, timeout=900)
    assert many.returncode == 0, many.stderr[-2000:]
    from _lines import printed_lines as pick
    assert len(pick(one.stdout)) >= 4 and pick(many.stdout) == pick(one.stdout), (one.stdout, many.stdout)
