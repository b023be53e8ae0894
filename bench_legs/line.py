"""The driver's line: the full record -> the compact last stdout line (<= 6 KB), and the side file with everything."""
import json
import os

from .common import ROOT

# ---- the driver's line ------------------------------------------------------------------------------------------------
MAX_LINE_BYTES = 6144


def _r(x, nd=4):
    """numbers to `nd` significant digits (the side file keeps full precision)"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    if isinstance(x, dict):
        return {k_: _r(v, nd) for k_, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def _pick(d, keys):
    return {k_: d[k_] for k_ in keys if isinstance(d, dict) and k_ in d}


def compact_line(full, detail_file=None):
    """The full record -> the line the driver parses: the contract's scalar fields verbatim, ONE `roofline` (dominant kernel family),
    a one-number-per-family table, and the headline numbers of the side measurements.  Everything else stays in the side file.
    Guaranteed <= MAX_LINE_BYTES: optional objects are dropped (largest first) if a future field ever pushes it over."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                        "dtype", "data", "config", "dist_per_s"))
    ro = full.get("roofline")
    if isinstance(ro, dict):
        line["roofline"] = _pick(ro, ("family", "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_unit",
                                      "traffic_over_algorithmic", "traffic_source", "share_of_step", "launches_per_step", "ms_per_step",
                                      "algorithmic_flop_per_step", "algorithmic_bytes_per_step", "frac_of_per_launch_rooflines", "timing", "hot_kernels",
                                      "traffic_profile"))
        if isinstance(line["roofline"].get("traffic_source"), str):
            line["roofline"]["traffic_source"] = line["roofline"]["traffic_source"].split(":")[0]
    else:
        line["roofline"] = None
    if isinstance(full.get("roofline_step"), dict):
        line["roofline_step"] = _pick(full["roofline_step"], ("bound", "achieved", "peak", "unit", "frac", "algorithmic_flop_per_step_per_gpu"))
    fams = {}
    for key, o in full.items():
        if key.startswith("roofline_") and key != "roofline_step" and isinstance(o, dict):
            fams[key[len("roofline_"):]] = [o.get("bound"), o.get("frac"), o.get("ms_per_step", o.get("launch_ms"))]
    if fams:
        line["families"] = {"columns": ["bound", "frac", "ms_per_step"], "rows": fams}
    sh = full.get("retrieval_shard")
    if isinstance(sh, dict):
        c = _pick(sh, ("error", "shape", "gallery_rows_per_gpu", "k", "ms", "dist_per_s", "tflops_end_to_end", "frac_of_f16_mfma_peak",
                       "identical_to_fp32_path", "includes"))
        if isinstance(sh.get("fp32_path"), dict):
            c["fp32_path"] = _pick(sh["fp32_path"], ("ms", "dist_per_s", "frac_of_f32_mfma_peak"))
        if isinstance(sh.get("sharded_average_precision"), dict):
            c["sharded_average_precision"] = _pick(sh["sharded_average_precision"], ("ms", "queries", "mAP"))
        line["retrieval_shard"] = c
    rg = full.get("extraction_regions")
    if isinstance(rg, dict):
        c = _pick(rg, ("error", "images_per_s", "ms_per_launch", "images_per_launch", "all_convolutions_in_libisx", "images_per_s_all_gpus", "note"))
        if isinstance(rg.get("roofline"), dict):
            c["roofline"] = _pick(rg["roofline"], ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic"))
        if isinstance(rg.get("retrieval_1000x100000"), dict):
            c["retrieval_1000x100000"] = _pick(rg["retrieval_1000x100000"], ("descriptor_dim", "total_ms", "dist_per_s",
                                                                              "cosine_sim_frac_of_f32_mfma_peak", "mAP"))
        c["workload"] = "BASELINE configs[2]: ResNet-50 TuneClassifSub @448x448 -> best-location descriptors"
        line["extraction_regions"] = c
    ig = full.get("ingest_streaming")
    if isinstance(ig, dict):
        line["ingest_streaming"] = _pick(ig, ("error", "images", "resident_images_per_s", "extract_pcie_inclusive_images_per_s", "streamed_over_resident",
                                              "descriptors_identical"))
    dg = full.get("ingest_decode")
    if isinstance(dg, dict):
        line["ingest_decode"] = _pick(dg, ("error", "images", "cores", "images_per_s", "decode_only_images_per_s", "decode_bound", "descriptors_identical_to_decode_first"))
    sl = full.get("slab_roundtrip")
    if isinstance(sl, dict):
        line["slab_roundtrip"] = _pick(sl, ("error", "rows", "write_GB_per_s", "read_GB_per_s", "identical", "search_identical"))
    tr = full.get("training")
    if isinstance(tr, dict):
        line["training"] = _pick(tr, ("error", "reference_config_triplets_per_s", "frozen_trunk_triplets_per_s", "reference_over_frozen", "reference_config", "statistic",
                                      "reference_config_with_prefix_cache_triplets_per_s"))
        if isinstance(tr.get("roofline"), dict):
            line["training"]["roofline"] = _pick(tr["roofline"], ("bound", "achieved", "peak", "unit", "frac", "ms_per_step", "algorithmic_flop_per_step"))
    if "exchange_ms" in full:
        line["exchange_ms"] = full["exchange_ms"]
        line["exchange"] = _pick(full.get("exchange") or {}, ("query_allgather_ms", "result_allgather_merge_ms", "exposed_when_serialised_frac_of_step",
                                                              "overlapped", "overlap_identical", "implementation", "communicators_in_data_path",
                                                              "merged_lists_identical_to_unsharded_search"))
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("error", "value", "unit", "cores", "kind", "sample", "cpu_model", "top1_matches_oracle_on_sample"))
        if isinstance(cb.get("retrieval"), dict):
            c["retrieval"] = _pick(cb["retrieval"], ("error", "value", "unit", "cores", "kind", "sample", "mm_tflops", "ap_loop_ms_per_query",
                                                      "ap_loop_sample", "ap_loop_error"))
        line["cpu_baseline"] = c
    if detail_file:
        line["detail_file"] = detail_file
    line = _r(line)
    for key in ("value", "ms_per_step", "dist_per_s"):          # the contract's scalars keep their digits
        if key in full:
            line[key] = full[key]
    for victim in ("families", "slab_roundtrip", "training", "ingest_decode", "ingest_streaming", "exchange", "extraction_regions", "retrieval_shard", "roofline_step"):      # never expected: a safety net
        if len(json.dumps(line)) <= MAX_LINE_BYTES:
            break
        line.pop(victim, None)
        line.setdefault("dropped_for_size", []).append(victim)
    return line


def write_detail(full):
    """The full record -> bench_detail.json (gpurun_out/ when present: that directory travels back from the GPU box)."""
    out_dir = os.path.join(ROOT, "gpurun_out")
    path = os.path.join(out_dir if os.path.isdir(out_dir) else ROOT, "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return os.path.relpath(path, ROOT)
    except Exception:
        return None
