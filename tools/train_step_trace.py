"""From a rocprofv3 --kernel-trace CSV of tools/bench_train.py: the launches of ONE steady training step (between two head_sgd_kernel launches in the
middle of the run), in order: short kernel name, grid, workgroup, microseconds.  usage: train_step_trace.py <kernel_trace.csv> [out.csv]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "head_sgd_kernel" in r["Kernel_Name"]]
a, b = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
t0 = int(rows[a]["End_Timestamp"])
out = []
for r in rows[a + 1:b + 1]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("isx::", "")
    out.append((name[:70], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1), int(r["Workgroup_Size_X"]),
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e3))
w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["kernel", "grid_threads", "wg", "us", "start_us_in_step"])
for o in out:
    w.writerow([o[0], o[1], o[2], "%.1f" % o[3], "%.1f" % o[4]])
busy = sum(o[3] for o in out)
wall = (int(rows[b]["End_Timestamp"]) - t0) / 1e3
w.writerow(["(sum of kernel time / wall)", "", "", "%.1f" % busy, "%.1f" % wall])
