#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the tracked summaries of profiles/.

  python profiles/summarize_prof.py <tag> <dir of `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`>
                                    [<dir of the --pmc FETCH_SIZE pass> <dir of the --pmc WRITE_SIZE pass>
                                     [<dir of the --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass>]]

Writes
  profiles/<tag>_kernel_stats.csv           every isx:: kernel + the 12 heaviest others (the rocprofv3 --stats table)
  profiles/<tag>_isx_kernels_by_shape.csv   libisx kernels averaged per launch shape (grid size) over the whole run
  profiles/<tag>_step_breakdown.csv         ONE steady-state bench step (the launches between two consecutive gap_l2 launches)
and, when the two PMC passes are given (same command, counters in passes of their own: FETCH_SIZE and WRITE_SIZE do not fit one pass),
  profiles/<tag>_pmc_hbm.csv                HBM bytes per kernel family of one steady-state step
  profiles/roofline_traffic.json            the same numbers keyed the way bench.py looks them up, stamped with their source
and, with the third counter pass,
  profiles/<tag>_pmc_mfma.csv               matrix-pipe utilisation per kernel family of one steady-state step: SQ_VALU_MFMA_BUSY_CYCLES /
                                            (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), the average clock (cycles / kernel time) and, from both,
                                            the TFLOP/s the family could reach at 100 % pipe utilisation AT THAT CLOCK (v_mfma_f32_32x32x2_f32:
                                            4096 FLOP per 64 cycles = 64 FLOP per SIMD and cycle; 157.3 TFLOP/s at 2.4 GHz)
HBM bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced
reads, WRITE_SIZE is exact; both count at the L2's fabric side, so Infinity-Cache hits are included)."""
import collections
import csv
import datetime
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits[-1]


def is_isx(name):
    return "isx::" in name or "_ZN3isx" in name


def family(name):
    """bench.py's roofline family of a kernel name (None: not a family the bench reports)."""
    if "conv1x1_stream_kernel" in name or "conv1x1_tail_kernel" in name:
        return "isx_conv1x1_nhwc"
    if "conv1x1_dual_nhwc_kernel" in name or "conv1x1_dual_tail_kernel" in name:
        return "isx_conv1x1_dual_nhwc"
    if "conv3x3_nhwc_kernel" in name or "conv3x3_tail_kernel" in name:
        return "isx_conv3x3_nhwc"
    if "conv3x3_expand_kernel" in name:
        return "isx_conv3x3_expand_nhwc"
    if "stem7x7_pool_kernel" in name:
        return "isx_stem7x7_pool_nhwc"
    if "gap_l2_nhwc_kernel" in name or "gap_l2_kernel" in name:
        return "gap_l2"
    if "cosine_gemm_kernel<" in name:
        args = name.split("cosine_gemm_kernel<", 1)[1].split(">", 1)[0].replace(" ", "").split(",")
        return "isx_conv1x1_nhwc" if args[3] == "2" else "cosine_gemm"
    return None


def one_step(rows, name_key, start_key):
    """The launches of one steady-state step: after the second-to-last gap_l2 launch up to (and including) the last one."""
    rows = sorted(rows, key=lambda r: int(r[start_key]))
    idx = [i for i, r in enumerate(rows) if "gap_l2" in r[name_key]]
    if len(idx) < 3:
        raise SystemExit("fewer than three steps in the trace")
    a, b = idx[-3], idx[-2]                        # not the very last step: the instrumented pass may follow different paths
    return rows[a + 1:b + 1]


def main():
    tag, run_dir = sys.argv[1], sys.argv[2]
    stats = find(run_dir, "_kernel_stats.csv")
    trace = find(run_dir, "_kernel_trace.csv")
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if is_isx(r["Name"])]
    others = [r for r in rows if not is_isx(r["Name"])][:12]
    with open(os.path.join(HERE, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep + others:
            r = dict(r)
            r["Name"] = r["Name"][:160]
            w.writerow(r)
    trows = list(csv.DictReader(open(trace)))
    per = collections.defaultdict(list)
    for r in trows:
        if is_isx(r["Kernel_Name"]):
            per[(r["Kernel_Name"].split("(")[0][:90], r["Grid_Size_X"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(HERE, tag + "_isx_kernels_by_shape.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_threads", "workgroup", "launches", "avg_us", "min_us", "max_us"])
        for (k, g, wg), v in sorted(per.items()):
            w.writerow([k, g, wg, len(v), "%.1f" % (sum(v) / len(v) / 1e3), "%.1f" % (min(v) / 1e3), "%.1f" % (max(v) / 1e3)])
    step = one_step(trows, "Kernel_Name", "Start_Timestamp")
    wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
    agg = collections.OrderedDict()
    for r in step:
        k = (r["Kernel_Name"].split("(")[0][:90], r["Grid_Size_X"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open(os.path.join(HERE, tag + "_step_breakdown.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_threads", "launches_in_step", "us_in_step", "share_of_step_wall", "family"])
        for (k, g), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, g, n, "%.1f" % us, "%.4f" % (us / wall), family(k) or ""])
        w.writerow(["(step wall clock, first launch start to last launch end)", "", len(step), "%.1f" % wall, "1.0", ""])
    if len(sys.argv) >= 5:
        fam = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
        for cname, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            crow = [r for r in csv.DictReader(open(find(d, "_counter_collection.csv"))) if r["Counter_Name"] == cname]
            for r in one_step(crow, "Kernel_Name", "Start_Timestamp"):
                fm = family(r["Kernel_Name"])
                if fm:
                    fam[fm][cname] += float(r["Counter_Value"])
                    if cname == "FETCH_SIZE":
                        fam[fm]["launches"] += 1
        out = {}
        with open(os.path.join(HERE, tag + "_pmc_hbm.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["family", "launches_in_step", "FETCH_SIZE_KiB_sum", "WRITE_SIZE_KiB_sum", "hbm_bytes_per_step_corrected(2*FETCH+WRITE)"])
            for fm, d in sorted(fam.items()):
                tot = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
                w.writerow([fm, d["launches"], "%.1f" % d["FETCH_SIZE"], "%.1f" % d["WRITE_SIZE"], "%.0f" % tot])
                out[fm] = {"bytes": tot, "launches": d["launches"]}
        src = ("profiles/%s_pmc_hbm.csv: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (kernel trace only) over one steady-state step "
               "of `python bench.py`, bytes = 2 x FETCH_SIZE + WRITE_SIZE, summed over the family's launches in the step; collected %s"
               % (tag, datetime.date.today().isoformat()))
        sys.path.insert(0, os.path.dirname(HERE))
        import bench                                  # csrc_digest(): the kernel sources these counters were read from
        json.dump({"source": src, "csrc_digest": bench.csrc_digest(), "kernels": out}, open(os.path.join(HERE, "roofline_traffic.json"), "w"), indent=1)
    if len(sys.argv) >= 6:
        rows = list(csv.DictReader(open(find(sys.argv[5], "_counter_collection.csv"))))
        one = {}
        for cname in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            one[cname] = one_step([r for r in rows if r["Counter_Name"] == cname], "Kernel_Name", "Start_Timestamp")
        fam = collections.defaultdict(lambda: collections.defaultdict(float))
        for cname, rs in one.items():
            for r in rs:
                fm = family(r["Kernel_Name"]) or "(other)"
                fam[fm][cname] += float(r["Counter_Value"])
                if cname == "GRBM_GUI_ACTIVE":
                    fam[fm]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    fam[fm]["launches"] += 1
        with open(os.path.join(HERE, tag + "_pmc_mfma.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["family", "launches_in_step", "kernel_us_in_step(with counters on)", "avg_clock_GHz", "mfma_pipe_busy", "fp32_mfma_peak_at_that_clock_TFLOPs"])
            for fm, d in sorted(fam.items()):
                cyc = d["GRBM_GUI_ACTIVE"] / 8.0
                if cyc <= 0 or d["ns"] <= 0:
                    continue
                clock = cyc / d["ns"]
                w.writerow([fm, int(d["launches"]), "%.1f" % (d["ns"] / 1e3), "%.3f" % clock, "%.3f" % (d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)),
                            "%.1f" % (clock * 1024 * 64 / 1e3)])


if __name__ == "__main__":
    main()
