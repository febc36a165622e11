#!/bin/bash
# round 3, GPU call O: the N > 1 code path of bench.py rehearsed on one GPU (2 ranks, gloo), and the gather with one rank over RCCL
mkdir -p gpurun_out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 50 --warmup 5 --rehearse-on-one-gpu --api rlbase > gpurun_out/r3o_rehearse2.json 2> gpurun_out/r3o_rehearse2.err; echo "rehearse rc=$?"; cut -c1-300 gpurun_out/r3o_rehearse2.json; tail -3 gpurun_out/r3o_rehearse2.err
timeout -k 10 300 python bench.py --gather --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/r3o_gather1.json 2> gpurun_out/r3o_gather1.err; echo "gather rc=$?"; python3 -c "
import json; d=json.loads(open('gpurun_out/r3o_gather1.json').read().strip().splitlines()[-1]); print(d['value'], d.get('gather'))"
