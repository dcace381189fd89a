import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 60 kernels before the final 200 (steady state region near the end of the timed loop)
sel = rows[-260:-200]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    print("%9.1f %8.1f q%s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
