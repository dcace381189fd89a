"""Kernel timeline of a rocprofv3 --kernel-trace run: python scripts/timeline.py <dir> [first-from-end] [count]
start (us), duration (us), gap to the previous kernel's end on the same queue (us), queue, kernel."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 260
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-back:-back + count]
t0 = int(sel[0]["Start_Timestamp"])
last_end = {}
for r in sel:
    q = r.get("Queue_Id", "?")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    print("%9.1f %8.1f %7.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, r["Kernel_Name"].replace("vc::", "").replace("void ", "")[:60]))
