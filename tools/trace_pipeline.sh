#!/bin/bash
# Runs on the GPU box: kernel trace of the timed steps of `bench.py --no-kernel-timing` and the timeline summary
# (tools/timeline_steps.py) + in-pipeline durations of the side kernels.   Usage: tools/trace_pipeline.sh <tag> [bench args]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p "$O"
rocprofv3 --kernel-trace --output-format csv -d $O/trace_tl -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-kernel-timing $* > $O/trace_tl.log 2>&1
python3 tools/timeline_steps.py $O/trace_tl > $O/timeline.txt 2>&1
tail -2 $O/timeline.txt
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace_tl/**/*_kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
lk = [r for r in rows if "k_lk" in r[2]]
t0 = lk[-34][0] if len(lk) > 40 else lk[0][0]
d = collections.defaultdict(list)
for s, e, n in rows:
    if s >= t0:
        k = n.replace("icelk::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        d[k].append((e - s) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-42s n=%3d mean %7.1f  median %7.1f  max %7.1f us" % (k, len(v), sum(v) / len(v), v2[len(v) // 2], v2[-1]))
PY
