for cfg in ref c5 c2; do
  for sb in 0 1; do
    ICELK_STAGE_BLOCK_AT=$sb timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg > gpurun_out/ab2_${cfg}_$sb.json 2> gpurun_out/ab2.err
    python -c "
import json;d=json.load(open('gpurun_out/ab2_${cfg}_$sb.json'));print('$cfg block_at $sb', round(d['value'],1), round(d['roofline']['avg_launch_us'],1))"
  done
done
