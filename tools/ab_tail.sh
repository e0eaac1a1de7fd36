#!/bin/bash
# same-box A/B of the detection tail: device-driven (default) against the host's (ICELK_HOST_TAIL=1), every configuration
for cfg in c2 c3 ref c5; do
  for mode in dev host; do
    if [ $mode = host ]; then export ICELK_HOST_TAIL=1; else unset ICELK_HOST_TAIL; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg > gpurun_out/ab_${cfg}_$mode.json 2> gpurun_out/ab_${cfg}_$mode.err || { echo "$cfg $mode failed"; tail -5 gpurun_out/ab_${cfg}_$mode.err; }
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab_${cfg}_$mode.json")); r=d["roofline"]
    print("$cfg $mode", round(d["value"],1), "lk/launch", round(r["avg_launch_us"],1), "pairs/launch", round(r.get("frame_pairs_per_launch"),2), "pcie", (d.get("pcie_inclusive") or {}).get("value"))
except Exception as e:
    print("$cfg $mode failed", e)
PY
  done
done
