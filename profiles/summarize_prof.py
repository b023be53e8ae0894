#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the tracked summaries of profiles/.

  python profiles/summarize_prof.py <round-tag> <kernel_stats.csv> [<pmc FETCH_SIZE csv> <pmc WRITE_SIZE csv>]

Writes profiles/<tag>_kernel_stats.csv (every isx:: kernel + the 12 heaviest others) and, when
PMC passes are given, profiles/<tag>_pmc_hbm.csv plus profiles/roofline_traffic.json (HBM bytes per
launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB: MI355X_MICROARCH.md, FETCH_SIZE reads 1/2 of wide
coalesced loads on gfx950, WRITE_SIZE is exact)."""
import collections
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(stats)))
    keep = [r for r in rows if "isx::" in r["Name"] or "_ZN3isx" in r["Name"]]
    others = [r for r in rows if "isx::" not in r["Name"] and "_ZN3isx" not in r["Name"]][:12]
    with open(os.path.join(HERE, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep + others:
            r = dict(r)
            r["Name"] = r["Name"][:160]
            w.writerow(r)
    # per-launch-shape averages of the hand-written kernels from the kernel trace next to the stats file
    trace = stats.replace("_kernel_stats.csv", "_kernel_trace.csv")
    if os.path.exists(trace):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(trace)):
            if "isx::" in r["Kernel_Name"] or "_ZN3isx" in r["Kernel_Name"]:
                per[(r["Kernel_Name"].split("(")[0][:90], r["Grid_Size_X"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        with open(os.path.join(HERE, tag + "_isx_kernels_by_shape.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "grid_threads", "workgroup", "launches", "avg_us", "min_us", "max_us"])
            for (k, g, wg), v in sorted(per.items()):
                w.writerow([k, g, wg, len(v), "%.1f" % (sum(v) / len(v) / 1e3), "%.1f" % (min(v) / 1e3), "%.1f" % (max(v) / 1e3)])
    if len(sys.argv) >= 5:
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for cname, path in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            for r in csv.DictReader(open(path)):
                if r["Counter_Name"] == cname and ("isx::" in r["Kernel_Name"] or "_ZN3isx" in r["Kernel_Name"]):
                    per[(r["Kernel_Name"][:100], r["Grid_Size"])][cname].append(float(r["Counter_Value"]))
        out = {}
        with open(os.path.join(HERE, tag + "_pmc_hbm.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "grid_size", "launches", "FETCH_SIZE_KiB_avg", "WRITE_SIZE_KiB_avg", "hbm_bytes_per_launch_corrected"])
            for (k, grid), d in sorted(per.items()):
                fe = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
                wr = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
                tot = (2.0 * fe + wr) * 1024.0
                w.writerow([k, grid, len(d["FETCH_SIZE"]), "%.1f" % fe, "%.1f" % wr, "%.0f" % tot])
                out.setdefault(k.split("(")[0].replace("void isx::", "").split("<")[0], {})[grid] = tot
        # bench.py looks the GEMM traffic up by problem shape: grid = tiles * 256 threads
        traffic = {"cosine_gemm_kernel": {}}
        for grid, tot in out.get("cosine_gemm_kernel", {}).items():
            tiles = int(grid) // 256
            # bench step 1024 x 10000: whichever tile shape the launcher picked (128x128, 64x128, 128x64, 64x64)
            for shape, ts in (("1024x10000x2048", (8 * 79, 16 * 79, 8 * 157, 16 * 157)), ("10000x32768x2048", (79 * 256,))):
                if tiles in ts:
                    traffic["cosine_gemm_kernel"][shape] = tot
        json.dump(traffic, open(os.path.join(HERE, "roofline_traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
