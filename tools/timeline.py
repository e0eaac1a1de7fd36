#!/usr/bin/env python3
"""GPU busy/idle summary from a rocprofv3 kernel_trace.csv: union of kernel intervals vs wall span."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))) for r in csv.DictReader(open(f))]
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = [r for r in rows if "k_synth" not in r[2]][skip:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.1f us, busy %.1f us (%.1f%%), kernels %d" % ((t1 - t0) / 1e3, busy / 1e3, 100.0 * busy / (t1 - t0), len(rows)))
if len(sys.argv) > 3:
    for s, e, n, q in rows[: int(sys.argv[3])]:
        print("%10.1f %8.1f  q=%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n.replace("icelk::(anonymous namespace)::", "")[:60]))
