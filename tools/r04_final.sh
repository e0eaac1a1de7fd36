#!/bin/bash
# Runs on the GPU box: the bench lines of every configuration as the round leaves them (with the CPU baseline where the driver
# would run it), the driver's own short run, a long soak, and the pyramid kernel alone (stamps + rocprofv3 stats)
cd "$GRAFT_REPO_ROOT"
for cfg in c2 c3 c4 ref c5; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --config $cfg > gpurun_out/r04_bench_$cfg.json 2> gpurun_out/cfg_$cfg.err || echo "$cfg failed"
done
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/default.err || echo "default failed"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_steps20.json 2> gpurun_out/driver.err || echo "driver failed"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --settle-ms 0 --no-cpu-baseline > gpurun_out/r04_bench_driver_steps20_no_settle.json 2>/dev/null || echo "driver (no settle) failed"
timeout -k 10 600 python bench.py --steps 4000 --warmup 20 --no-cpu-baseline > gpurun_out/r04_bench_soak_4000steps.json 2>/dev/null || echo "soak failed"
python - <<'PY'
import json
for n in ("c2", "c3", "c4", "ref", "c5", "default", "driver_steps20", "driver_steps20_no_settle", "soak_4000steps"):
    try:
        d = json.load(open("gpurun_out/r04_bench_%s.json" % n)); k = d.get("kernel_rooflines", {})
        print("%-26s %8.1f pairs/s  steps %d  lk/launch %.1f alone %.1f  pyramid alone %s  pcie %s  cpu %s" % (
            n, d["value"], d["steps"], d["roofline"]["avg_launch_us"], k.get("lk_fb", {}).get("alone_us", 0),
            k.get("pyramid", {}).get("alone_us_per_frame"), (d.get("pcie_inclusive") or {}).get("value"),
            (d.get("cpu_baseline") or {}).get("value")))
    except Exception as e:
        print(n, "unreadable", e)
PY
bash tools/pyr_stamps_run.sh > gpurun_out/r04_pyramid_stamps.txt 2>&1
bash tools/pyr_profile.sh >> gpurun_out/r04_pyramid_stamps.txt 2>&1
python3 tools/pyr_stamps_map.py gpurun_out/pyr_stamps_wide.bin 32 24 >> gpurun_out/r04_pyramid_stamps.txt 2>&1
cp $(ls -t gpurun_out/pyr_wide/*/*_kernel_stats.csv | head -1) gpurun_out/r04_pyramid_alone_kernel_stats.csv
cp $(ls -t gpurun_out/pyr_1w/*/*_kernel_stats.csv | head -1) gpurun_out/r04_pyramid_one_wave_alone_kernel_stats.csv
grep -v '^"void' gpurun_out/r04_pyramid_stamps.txt | tail -14
grep -h pyramid gpurun_out/r04_pyramid_alone_kernel_stats.csv gpurun_out/r04_pyramid_one_wave_alone_kernel_stats.csv | cut -c200-330
