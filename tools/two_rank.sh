#!/bin/bash
# Two ranks of bench.py on the ONE GPU of a gpurun box: rehearses the sharded sequence (frame blocks per rank, segment
# archive on the device, gathers) end to end.  Both ranks share device 0, where an RCCL communicator cannot be formed
# ("Duplicate GPU detected"), so the RCCL leg is skipped explicitly (ICELK_BENCH_SHARED_DEVICE) and the gathers go over
# gloo; on a real multi-GPU node the same script without that variable exercises RCCL.
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 WORLD_SIZE=2 LOCAL_RANK=0 ICELK_BENCH_SHARED_DEVICE=1
RANK=1 timeout -k 10 300 python bench.py --gpus 2 --steps 40 --warmup 4 > gpurun_out/r1.out 2> gpurun_out/r1.err &
P1=$!
RANK=0 timeout -k 10 300 python bench.py --gpus 2 --steps 40 --warmup 4 > gpurun_out/r0.out 2> gpurun_out/r0.err
echo "rank0 rc=$?"
wait $P1
echo "rank1 rc=$?"
wc -l gpurun_out/r0.out gpurun_out/r1.out
python -c "
import json; d=json.load(open('gpurun_out/r0.out')); print(d['n_gpus'], round(d['value'],1), d['config']['workload'][:40], d['gather'])"
tail -2 gpurun_out/r0.err | cut -c1-300
