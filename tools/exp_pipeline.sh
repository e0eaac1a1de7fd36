run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/x_$name.json 2> gpurun_out/x_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/x_$name.json')); print('$name', round(d['value'],1), 'lk', round(d['roofline']['avg_launch_us'],1))"
}
run probe ICELK_STREAM_PROBE_LOG=1
grep "icelk probe" gpurun_out/x_probe.err | head -40
run noprobe ICELK_NO_STREAM_PROBE=1
run probe2 A=1
