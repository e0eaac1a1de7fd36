export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 WORLD_SIZE=2 LOCAL_RANK=0
RANK=1 timeout -k 10 240 python bench.py --gpus 2 --steps 40 --warmup 4 --ring 8 > gpurun_out/r1.out 2> gpurun_out/r1.err &
P1=$!
RANK=0 timeout -k 10 240 python bench.py --gpus 2 --steps 40 --warmup 4 --ring 8 > gpurun_out/r0.out 2> gpurun_out/r0.err
wait $P1
echo "rank1 rc=$?"
wc -l gpurun_out/r0.out gpurun_out/r1.out
python -c "
import json; d=json.load(open('gpurun_out/r0.out')); print(d['n_gpus'], round(d['value'],1), d['config']['count_gather'], d['tracked_features_per_sec'])"
tail -2 gpurun_out/r0.err | cut -c1-300
