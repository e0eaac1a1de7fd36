#!/bin/bash
# Runs on the GPU box: a 20-step window behind 5, 200, 1000 and 3000 warm-up steps
for kw in "20 5" "20 200" "20 1000" "20 3000" "20 5"; do
  set -- $kw
  for r in 1 2 3; do
    python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-kernel-timing > gpurun_out/_sw.json 2>/dev/null || { echo failed; exit 1; }
    python -c "
import json;d=json.load(open('gpurun_out/_sw.json'))
print('steps $1 warmup $2: %.0f pairs/s  %.1f us/step  window %.0f us' % (d['value'], 1e3*d['ms_per_step'], 1e3*d['ms_per_step']*$1))"
  done
done
