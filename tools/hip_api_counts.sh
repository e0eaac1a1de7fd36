#!/bin/bash
# Runs on the GPU box: HIP runtime calls of the resident loop per step (hipEventRecord / hipStreamWaitEvent / launches ...):
# the difference between a 400-step and a 200-step run, divided by 200.   Usage: tools/hip_api_counts.sh <tag>
TAG=${1:-hipapi}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for K in 200 400; do
  O=gpurun_out/$TAG/k$K; rm -rf $O; mkdir -p $O
  rocprofv3 --hip-runtime-trace --stats --output-format csv -d $O -- python3 bench.py --steps $K --warmup 20 --settle-ms 0 --no-cpu-baseline --no-kernel-timing > $O/log.txt 2>&1
done
python3 - gpurun_out/$TAG <<'PY'
import csv, glob, sys
def load(d):
    f = glob.glob(d + "/**/*hip_api_stats.csv", recursive=True) or glob.glob(d + "/**/*_stats.csv", recursive=True)
    out = {}
    for r in csv.DictReader(open(f[0])):
        out[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return out
a, b = load(sys.argv[1] + "/k200"), load(sys.argv[1] + "/k400")
print("%-34s %10s %12s" % ("HIP call", "per step", "host us/step"))
tot = 0
for n in sorted(b, key=lambda n: -(b[n][1] - a.get(n, (0, 0))[1])):
    dc, dt = b[n][0] - a.get(n, (0, 0))[0], b[n][1] - a.get(n, (0, 0))[1]
    if dc:
        print("%-34s %10.2f %12.2f" % (n, dc / 200.0, dt / 200e3)); tot += dt / 200e3
print("host time inside HIP calls: %.1f us per step" % tot)
PY
