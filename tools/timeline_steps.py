#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `bench.py --no-kernel-timing` (no per-kernel "alone" launches behind the timed region:
the last tracker launches of the trace ARE timed steps): the kernels of a few steady-state steps, one line each (start
relative to the first tracker launch shown, duration, queue), then how much of the steady-state span the tracker launches
cover and the gaps between consecutive tracker launches."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f))]
rows.sort()
def short(n):
    return n.replace("icelk::(anonymous namespace)::", "").replace("icelk::", "").replace("void ", "").split("(")[0][:44]
lk = [i for i, r in enumerate(rows) if "k_lk" in r[2]]
if len(lk) < 16:
    print("too few tracker launches", len(lk)); sys.exit(0)
a, b = lk[-7], lk[-4]      # three tracker launches
t0 = rows[a][0]
for s, e, n, q in rows[a:b + 1]:
    print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q[-3:], short(n)))
# steady state: the last 16 tracker launches
s0, s1 = rows[lk[-17]][0], rows[lk[-1]][0]
lkt = sum(rows[i][1] - rows[i][0] for i in lk[-17:-1])
print("steady state: %.1f us per tracker launch period, tracker kernels cover %.1f%% of it" % ((s1 - s0) / 16e3, 100.0 * lkt / (s1 - s0)))
gaps = [(rows[lk[k + 1]][0] - rows[lk[k]][1]) / 1e3 for k in range(len(lk) - 17, len(lk) - 1)]
print("gap between consecutive tracker launches (us):", " ".join("%.0f" % g for g in gaps))
