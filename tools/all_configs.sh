#!/bin/bash
# every bench configuration once on the one GPU of a gpurun box (no CPU baseline): value, tracker launch time, pipeline share
for cfg in c2 c3 c4 ref c5; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg > gpurun_out/cfg_$cfg.json 2> gpurun_out/cfg_$cfg.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/cfg_$cfg.json")); k=d["kernels"]
    r=d["roofline"]
    print("$cfg", round(d["value"],1), d["unit"], "lk/launch", round(r["avg_launch_us"],1), "pairs/launch", r.get("frame_pairs_per_launch"), "alone", d["kernel_rooflines"].get("lk_fb",{}).get("alone_us"), "pcie", (d.get("pcie_inclusive") or {}).get("value"))
except Exception as e:
    print("$cfg failed", e)
PY
done
