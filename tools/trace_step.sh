#!/bin/bash
# Runs on the GPU box: kernel trace of a short bench run, then the timeline of a few steady-state steps.
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-kernel-timing $* > $O/trace.log 2>&1
python3 tools/timeline_steps.py $O/trace > $O/timeline.txt 2>&1
cp $O/trace/*/*_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
cat $O/timeline.txt | head -90
