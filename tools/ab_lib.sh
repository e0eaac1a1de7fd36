#!/bin/bash
# same-box A/B of library builds (tools/build_variant.sh): usage tools/ab_lib.sh <config> <name|default> ...
cfg=$1; shift
for name in "$@"; do
  if [ $name = default ]; then unset ICELK_LIBRARY; else export ICELK_LIBRARY=$PWD/variants/libicelk_$name.so; fi
  for rep in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg > gpurun_out/abl_${cfg}_$name.json 2> gpurun_out/abl.err || { echo "$cfg $name failed"; tail -3 gpurun_out/abl.err; }
  python -c "
import json;d=json.load(open('gpurun_out/abl_${cfg}_$name.json'));print('$cfg $name: %.1f pairs/s, tracker launch %.1f us in the pipeline, %.1f us per pair alone' % (d['value'], d['roofline']['avg_launch_us'], d['kernel_rooflines']['lk_fb']['alone_us']))"
  done
done
