#!/bin/bash
# Runs on the GPU box: kernel trace of the driver's short run (bench.py --steps 20 --warmup 5): every tracker launch with
# the gap before it, and what ran in the longest gaps.   Usage: tools/trace_short.sh <tag>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p "$O"
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing $* > $O/trace.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
short = lambda n: n.replace("icelk::(anonymous namespace)::", "").replace("icelk::", "").replace("void ", "").split("(")[0][:40]
lk = [i for i, r in enumerate(rows) if "k_lk" in r[2]]
t0 = rows[lk[0]][0]
prev_end = None
for i in lk:
    s, e, n = rows[i]
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print("%9.1f us  %-28s %7.1f us   gap before %7.1f us" % ((s - t0) / 1e3, short(n), (e - s) / 1e3, gap))
    if gap > 60:
        for s2, e2, n2 in rows:
            if s2 >= prev_end - 1000 and s2 < s and "k_lk" not in n2:
                print("              in the gap: +%7.1f  %-34s %6.1f us" % ((s2 - prev_end) / 1e3, short(n2), (e2 - s2) / 1e3))
    prev_end = e
PY
