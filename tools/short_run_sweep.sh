#!/bin/bash
# Runs on the GPU box: what a 20-step timed window (the driver's) measures with and without the settle phase before the warm-up
# steps, beside longer windows.  Usage: tools/short_run_sweep.sh
run() {
  python bench.py --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/_sw.json 2>/dev/null || { echo failed; exit 1; }
  python -c "
import json,sys;d=json.load(open('gpurun_out/_sw.json'))
print('%-44s %.0f pairs/s  %.1f us/step  settle %.0f ms / %d steps' % (' '.join(sys.argv[1:]), d['value'], 1e3*d['ms_per_step'], d['settle']['ms'], d['settle']['steps']))" "$@"
}
for r in 1 2 3; do run --steps 20 --warmup 5 --settle-ms 0; done
for r in 1 2 3; do run --steps 20 --warmup 5; done
for r in 1 2; do run --steps 20 --warmup 5 --settle-ms 50; done
for r in 1 2; do run --steps 20 --warmup 5 --settle-ms 400; done
for r in 1 2; do run --steps 200 --warmup 20 --settle-ms 0; done
for r in 1 2; do run --steps 200 --warmup 20; done
for r in 1 2; do run --config ref --steps 20 --warmup 5 --settle-ms 0; done
for r in 1 2; do run --config ref --steps 20 --warmup 5; done
for r in 1 2; do run --config c3 --settle-ms 0; done
for r in 1 2; do run --config c3; done
