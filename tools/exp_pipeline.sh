run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/x_$name.json 2> gpurun_out/x_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/x_$name.json')); k=d['kernels']; print('$name', round(d['value'],1), 'lk', round(d['roofline']['avg_launch_us'],1), 'eig', k['corner_candidates']['avg_us'], 'mind', k['min_distance']['avg_us'], 'pyr', k['pyrdown']['avg_us'])"
}
run tail34 A=1
run tail23 ICELK_DET_AHEAD=2,3
run notail34 ICELK_NO_TAIL_STREAM=1
run tail34b A=1
run tail23b ICELK_DET_AHEAD=2,3
