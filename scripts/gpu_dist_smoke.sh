#!/bin/bash
# Functional check of the N>1 bench path on a 1-GPU box: 2 ranks share cuda:0, gloo backend (RCCL needs distinct GPUs).
export TINYFACES_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-profile > gpurun_out/bench_dist2.json 2> gpurun_out/bench_dist2.err
echo "dist exit $?"; tail -3 gpurun_out/bench_dist2.err; cat gpurun_out/bench_dist2.json
