#!/bin/bash
# per-stage clock stamps of the pyramid kernel (both geometries), alone on the device
for mode in wide 1w; do
  if [ $mode = 1w ]; then export PYR_AHEAD=1; else unset PYR_AHEAD; fi
  ICELK_PYR_STAMPS=gpurun_out/pyr_stamps_$mode.bin python3 tools/pyr_alone.py > /dev/null 2>&1
  echo $mode; python3 tools/pyr_stamps.py gpurun_out/pyr_stamps_$mode.bin
  python3 - <<PY
import numpy as np
a=np.fromfile("gpurun_out/pyr_stamps_$mode.bin",dtype=np.uint64).reshape(-1,8).astype(np.int64)
t0=a[:,0].min()
print("first start .. last end: %d cycles of s_memtime (100 MHz?) ; starts spread %d ; ends spread %d" % (a[:,5].max()-t0, a[:,0].max()-t0, a[:,5].max()-a[:,5].min()))
PY
done
