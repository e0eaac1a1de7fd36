#!/bin/bash
# same-box A/B of the corner kernel's workgroup shape beside the tracker: one-wave strips (default for candidates ahead)
# against round 3's four-wave strips (ICELK_STRIP_WAVES=4)
for cfg in c2 c3 ref c5; do
  for mode in 1w 4w; do
    if [ $mode = 4w ]; then export ICELK_STRIP_WAVES=4; else unset ICELK_STRIP_WAVES; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg > gpurun_out/abs_${cfg}_$mode.json 2> gpurun_out/abs_${cfg}_$mode.err || { echo "$cfg $mode failed"; tail -5 gpurun_out/abs_${cfg}_$mode.err; }
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/abs_${cfg}_$mode.json")); r=d["roofline"]; k=d["kernel_rooflines"]
    print("$cfg $mode", round(d["value"],1), "lk/launch", round(r["avg_launch_us"],1), "eig alone", round(k["corner_candidates"].get("alone_us",0),1), "pcie", (d.get("pcie_inclusive") or {}).get("value"))
except Exception as e:
    print("$cfg $mode failed", e)
PY
  done
done
