#!/bin/bash
# Runs on the GPU box: bench.py several times, one line per run (value, pairs launched, tracker launch us in the pipeline and alone,
# corner kernel alone).   Usage: tools/bench_repeat.sh <n> <steps> <warmup> [bench args / env in front via `env`]
N=$1; K=$2; W=$3; shift 3
for i in $(seq $N); do
  python bench.py --steps $K --warmup $W --no-cpu-baseline $* > gpurun_out/_rep.json 2>/dev/null || { echo "bench failed"; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/_rep.json'));k=d.get('kernel_rooflines',{})
print('steps $K: %.0f pairs/s  launched %s  lk in-pipeline %.1f us alone %.1f us  corner alone %.1f us' % (d['value'], d['pairs_launched_in_timed_region'], k.get('lk_fb',{}).get('avg_launch_us',0), k.get('lk_fb',{}).get('alone_us',0), k.get('corner_candidates',{}).get('alone_us',0)))"
done
