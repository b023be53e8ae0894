"""bench.py's last stdout line is what the driver parses: it must stay small (<= 6 KB; the round-3 line grew to 20.7 KB and the
driver's record lost it) and must survive json round trips, for N = 1 and for the N > 1 shape of the record (exchange fields).
CPU only: the full records are the ones committed under profiles/ by GPU runs of bench.py."""
import argparse
import glob
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("isx_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _full_records():
    recs = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_line.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_detail*.json"))):
        d = json.load(open(path))
        recs.append((os.path.basename(path), d.get("bench_detail", d)))
    return recs


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline")


@pytest.mark.parametrize("name,full", _full_records())
def test_compact_line_n1(name, full):
    b = _bench()
    line = b.compact_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) <= b.MAX_LINE_BYTES == 6144, (name, len(text))
    back = json.loads(text)
    assert back == line and "dropped_for_size" not in back
    for key in CONTRACT:
        assert key in back, key
    assert "workload" in back["config"] and "model" not in back["config"]
    r = back["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 * max(r["frac"], 1e-9) + 1e-6
    assert "traffic" in r
    if isinstance(full.get("cpu_baseline"), dict) and "value" in full["cpu_baseline"]:
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in back["cpu_baseline"], key
    # the contract's scalars are copied, not rounded away
    assert abs(back["value"] - full["value"]) <= 1e-3 * full["value"]
    assert abs(back["ms_per_step"] - full["ms_per_step"]) <= 1e-3 * full["ms_per_step"]


def test_compact_line_n2_shape():
    """The N > 1 record (ISX_BENCH_ONE_DEVICE=1 N = 2 path): exchange fields, ranks, backend; same size bound."""
    b = _bench()
    recs = _full_records()
    assert recs, "no committed full bench record under profiles/"
    full = json.loads(json.dumps(recs[-1][1]))
    full["n_gpus"] = 2
    full["config"] = dict(full["config"], ranks=2, collective_backend="nccl", parallelism="gallery-row shards x2 + DP extraction")
    full["exchange_ms"] = 0.4321
    full["exchange"] = {"query_allgather_ms": 0.21, "result_allgather_merge_ms": 0.22, "exposed_when_serialised_frac_of_step": 0.0066,
                        "overlapped": True, "overlap_identical": True, "implementation": "isx_shard_topk_allgather (grouped ncclAllGather x 2) + isx_topk_merge",
                        "legs": "x" * 400, "timing": "y" * 200}
    full["cpu_baseline"] = dict(full.get("cpu_baseline") or {"value": 27.8, "unit": "images/s", "cores": 16, "kind": "port", "sample": "s"},
                                retrieval={"value": 1.1e9, "unit": "distances/s", "cores": 16, "kind": "port", "sample": "z" * 160,
                                           "mm_tflops": 2.3, "ap_loop_ms_per_query": 15.9, "ap_loop_sample": "w" * 100})
    line = b.compact_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) <= 6144, len(text)
    back = json.loads(text)
    assert back["n_gpus"] == 2 and back["config"]["ranks"] == 2 and back["config"]["collective_backend"] == "nccl"
    assert back["exchange_ms"] == pytest.approx(0.4321, rel=1e-3) and back["exchange"]["overlap_identical"] is True
    assert back["cpu_baseline"]["retrieval"]["unit"] == "distances/s" and "dropped_for_size" not in back


def test_hot_kernels_ride_inside_roofline_and_stale_traffic_is_null():
    """north_star's named kernels (distance matmul + top-k, pooling) are reported INSIDE `roofline` -- the object the driver stores whole -- and
    HBM traffic from a PMC profile of OTHER kernel sources is reported as null, not as bytes."""
    b = _bench()
    full = json.loads(json.dumps(_full_records()[-1][1]))
    full["roofline"]["hot_kernels"] = {
        "cosine_topk_fp32": {"shape": [10000, 125000, 2048], "k": 100, "ms": 36.7, "tflops": 139.5, "peak": 157.3, "frac": 0.887, "bound": "mfma", "what": "w" * 110},
        "cosine_topk_fast": {"ms": 6.24, "identical": True, "frac_of_f16_peak": 0.33},
        "cosine_gemm_step": {"shape": [1024, 10000, 2048], "launch_ms": 0.32, "achieved": 131.0, "frac": 0.83},
        "gap_l2": {"GB_s": 5450.0, "frac": 0.68, "ms": 0.077, "bytes_per_image": 409600.0, "traffic": None}}
    full["roofline"]["traffic_profile"] = {"csrc_digest_of_profile": "0" * 16, "csrc_digest_now": "1" * 16, "fresh": False}
    line = b.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(json.dumps(line)) <= 6144 and "dropped_for_size" not in line
    hk = line["roofline"]["hot_kernels"]
    assert hk["cosine_topk_fp32"]["frac"] == pytest.approx(0.887) and hk["cosine_topk_fp32"]["shape"] == [10000, 125000, 2048]
    assert hk["cosine_topk_fast"]["identical"] is True and hk["gap_l2"]["frac"] == pytest.approx(0.68)
    assert line["roofline"]["traffic_profile"]["fresh"] is False
    # the digest is a function of the kernel sources only, and load_traffic() refuses a profile of other sources
    d = b.csrc_digest()
    assert len(d) == 16 and d == b.csrc_digest()
    t = b.load_traffic()
    prof = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))
    if prof.get("csrc_digest") == d:
        assert t["fresh"] and t["kernels"]
    else:
        assert t["fresh"] is False and t["kernels"] == {} and t["regions_leg"] == {}


def test_compact_line_safety_net_drops_optional_objects():
    b = _bench()
    full = json.loads(json.dumps(_full_records()[-1][1]))
    full["extraction_regions"] = dict(full.get("extraction_regions") or {}, note="n" * 9000)
    line = b.compact_line(full)
    assert len(json.dumps(line)) <= 6144 and "extraction_regions" in line["dropped_for_size"]
    assert line["value"] == pytest.approx(full["value"], rel=1e-3)


def test_cpu_baseline_retrieval_runs_on_the_host():
    """The retrieval half of cpu_baseline (torch.mm + topk + the literal AP loop): a short run of the same code bench.py times."""
    b = _bench()
    out = b.cpu_baseline_retrieval(argparse.Namespace(k=100), seconds=0.2)
    assert out["unit"] == "distances/s" and out["value"] > 1e6 and out["cores"] >= 1 and out["kind"] == "port"
    assert out["ap_loop_ms_per_query"] > 0 and "ap_loop_error" not in out
