#!/bin/bash
# Runs on the GPU box: the C3 configuration N times, one line per run: resident rate, PCIe-inclusive rate, what the link gave
# this process for its pinned buffers alone, and where process, buffers and device sit (NUMA).  Usage: tools/c3_repeat.sh <n>
N=${1:-8}
for i in $(seq $N); do
  timeout -k 10 300 python bench.py --config c3 --no-cpu-baseline > gpurun_out/_c3.json 2>gpurun_out/_c3.err || { echo "bench failed"; tail -3 gpurun_out/_c3.err; exit 1; }
  python -c "
import json;d=json.load(open('gpurun_out/_c3.json'));p=d['pcie_inclusive']
print('c3 run $i: resident %.0f  pcie-inclusive %.0f  raw upload %.1f us = %.1f GB/s  gpu node %s  host us/frame %s' % (d['value'], p['value'], p['raw_upload_us'], p['raw_upload_GBps'], p['numa'].get('gpu_node'), p['host_us_per_frame']))"
done
