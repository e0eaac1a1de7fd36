#!/bin/bash
# Two ranks of bench.py on the ONE GPU of a gpurun box, started the way the driver starts a multi-GPU run
# (`python bench.py --gpus 2 ...`: bench.py spawns its own ranks): rehearses the sharded sequence (frame blocks per rank,
# segment archive on the device, gathers) end to end.  Both ranks share device 0, where an RCCL communicator cannot be
# formed ("Duplicate GPU detected"), so the RCCL leg is skipped explicitly (ICELK_BENCH_SHARED_DEVICE) and the gathers go
# over gloo; on a real multi-GPU node the same command without that variable exercises RCCL.
export ICELK_BENCH_SHARED_DEVICE=1
timeout -k 10 400 python bench.py --gpus 2 --steps 40 --warmup 4 > gpurun_out/r0.out 2> gpurun_out/r0.err
echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r0.out')); print(d['n_gpus'], round(d['value'],1), d['config']['workload'][:40], d['gather'])"
tail -2 gpurun_out/r0.err | cut -c1-300
