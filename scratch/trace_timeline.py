"""Timeline of the last search in a rocprofv3 --kernel-trace of scratch/prof_fast.py: start / end (us, relative) per kernel and stream."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last search = from the last amax_kernel on
start = max(i for i, r in enumerate(rows) if "amax_kernel" in r["Kernel_Name"])
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    n = r["Kernel_Name"].split("(")[0][-46:]
    print("%9.1f -> %9.1f  %8.1f us  q=%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), n))
