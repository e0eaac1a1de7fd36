#!/bin/bash
# the rate of the pipeline under the "lk_sums" variants (icelk_set_variant: OpenCV's x86 float-lane accumulation order) beside
# the default (exact sums): REF and C2
for v in 0 1 2; do for cfg in ref c2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg --lk-sums $v > gpurun_out/r4_sums_${cfg}_$v.json 2>gpurun_out/r4_sums.err
  python -c "
import json;d=json.load(open('gpurun_out/r4_sums_${cfg}_$v.json'));print('$cfg lk_sums $v: %.1f pairs/s, tracker launch %.1f us in the pipeline, %.1f us per pair alone' % (d['value'], d['roofline']['avg_launch_us'], d['kernel_rooflines']['lk_fb']['alone_us']))"
done; done
