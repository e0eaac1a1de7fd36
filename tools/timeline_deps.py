#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: for each of the last tracker launches, when it started, what the device was
doing in the gap before it (kernels that ended inside the gap) and the kernels that overlap the launch."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
def short(n):
    return n.replace("icelk::(anonymous namespace)::", "").replace("icelk::", "").replace("void ", "").split("(")[0][:28]
lk = [i for i, r in enumerate(rows) if "k_lk" in r[2]]
sel = lk[-skip - 5:-skip]
t0 = rows[sel[0]][0]
for k, i in enumerate(sel[1:], 1):
    ps, pe = rows[sel[k - 1]][0], rows[sel[k - 1]][1]
    s, e = rows[i][0], rows[i][1]
    print("tracker %8.1f .. %8.1f (%.1f us)   previous ended %8.1f, gap %.1f" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (pe - t0) / 1e3, (s - pe) / 1e3))
    for r in rows:
        if r[1] > pe - 30000 and r[1] <= s + 2000 and "k_lk" not in r[2]:
            print("      %-28s q%-2s %8.1f .. %8.1f" % (short(r[2]), r[3][-2:], (r[0] - t0) / 1e3, (r[1] - t0) / 1e3))
