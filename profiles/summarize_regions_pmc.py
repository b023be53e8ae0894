#!/usr/bin/env python3
"""HBM bytes of ONE launch of the BASELINE configs[2] leg (`python bench.py --only-regions`) from two rocprofv3 counter passes
(--pmc FETCH_SIZE, --pmc WRITE_SIZE; kernel trace only): the kernels between two consecutive best_location_desc_nhwc_kernel launches.

  python profiles/summarize_regions_pmc.py <tag> <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [images per launch]

Writes profiles/<tag>_regions_pmc_hbm.csv (per kernel family) and merges {"regions_leg": {...}} into profiles/roofline_traffic.json, where
bench.py looks it up for extraction_regions.roofline.traffic.  bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (gfx950 correction, see summarize_prof.py)."""
import collections
import csv
import datetime
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_prof import HERE, family, find, is_isx  # noqa: E402


def one_launch(rows):
    rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "best_location_desc_nhwc_kernel" in r["Kernel_Name"]]
    if len(idx) < 4:
        raise SystemExit("fewer than four launches of the region leg in the trace")
    return rows[idx[-3] + 1:idx[-2] + 1]


def fam_of(name):
    if "boxpool_s1_nhwc_kernel" in name or "best_location_desc_nhwc_kernel" in name:
        return "region_tail"
    return family(name) or ("other_isx" if is_isx(name) else "other")


def main():
    tag, fdir, wdir = sys.argv[1], sys.argv[2], sys.argv[3]
    images = int(sys.argv[4]) if len(sys.argv) > 4 else 128
    fam = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
    for cname, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
        crow = [r for r in csv.DictReader(open(find(d, "_counter_collection.csv"))) if r["Counter_Name"] == cname]
        for r in one_launch(crow):
            f = fam_of(r["Kernel_Name"])
            fam[f][cname] += float(r["Counter_Value"])
            if cname == "FETCH_SIZE":
                fam[f]["launches"] += 1
    total = 0.0
    with open(os.path.join(HERE, tag + "_regions_pmc_hbm.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["family", "kernel_launches", "FETCH_SIZE_KiB_sum", "WRITE_SIZE_KiB_sum", "hbm_bytes_corrected(2*FETCH+WRITE)"])
        for k, d in sorted(fam.items()):
            b = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
            total += b
            w.writerow([k, d["launches"], "%.1f" % d["FETCH_SIZE"], "%.1f" % d["WRITE_SIZE"], "%.0f" % b])
        w.writerow(["(one launch of the leg: %d images of 448 x 448)" % images, "", "", "", "%.0f" % total])
    path = os.path.join(HERE, "roofline_traffic.json")
    try:
        doc = json.load(open(path))
    except Exception:
        doc = {}
    doc["regions_leg"] = {"bytes_per_launch": total, "images_per_launch": images,
                          "source": "profiles/%s_regions_pmc_hbm.csv: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `python bench.py --only-regions`, one launch "
                                    "of the leg, bytes = 2 x FETCH_SIZE + WRITE_SIZE; collected %s" % (tag, datetime.date.today().isoformat())}
    json.dump(doc, open(path, "w"), indent=1)
    print("regions leg: %.2f GB per launch of %d images" % (total / 1e9, images))


if __name__ == "__main__":
    main()
