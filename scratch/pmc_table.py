"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (+ launch time from the kernel trace beside it).
usage: python scratch/pmc_table.py <dir with *_counter_collection.csv>"""
import collections, csv, glob, sys
d = sys.argv[1]
cc = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    key = (r["Kernel_Name"][:70], r["Grid_Size"])
    per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Dispatch_Id"] in dur:
        per[key]["_us"].append(dur[r["Dispatch_Id"]])
for key, c in per.items():
    n = len(c["_us"]) // max(1, len(c) - 1) if "_us" in c else 0
    print(key[0], "grid", key[1])
    avg = {k: sum(v) / len(v) for k, v in c.items()}
    us = avg.get("_us", 0.0)
    print("   us %.1f" % us, " ".join("%s=%.4g" % (k, v) for k, v in sorted(avg.items()) if k != "_us"))
    if "GRBM_GUI_ACTIVE" in avg and us:
        clk = avg["GRBM_GUI_ACTIVE"] / 8 / (us * 1e-6) / 1e9
        line = "   clock %.3f GHz" % clk
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            line += "  mfma busy %.3f" % (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (avg["GRBM_GUI_ACTIVE"] / 8))
        if "SQ_LDS_IDX_ACTIVE" in avg:
            line += "  lds active/CU %.3f" % (avg["SQ_LDS_IDX_ACTIVE"] / 256 / (avg["GRBM_GUI_ACTIVE"] / 8))
        print(line)
