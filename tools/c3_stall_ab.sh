#!/bin/bash
# Runs on the GPU box: how often the PCIe-fed loop loses ~9 ms in one seg_detect_stage (tools/c3_stall.py), by the way the host waits
N=${1:-24}
for mode in spin event spin event; do
  if [ $mode = event ]; then export ICELK_EVENT_WAIT=1; else unset ICELK_EVENT_WAIT; fi
  timeout -k 10 400 python3 tools/c3_stall.py $N > gpurun_out/_stall_$mode.txt 2>&1
  echo "$mode: $(grep -c 'pairs/s' gpurun_out/_stall_$mode.txt) runs, stalled (< 3000 pairs/s): $(awk '/pairs\/s/ && $3 < 3000' gpurun_out/_stall_$mode.txt | wc -l)"
  awk '/pairs\/s/ && $3 < 3000' gpurun_out/_stall_$mode.txt | cut -c1-160
done
