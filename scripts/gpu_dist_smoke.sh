#!/bin/bash
# Functional checks of the data-parallel path on a 1-GPU box.
# (1) RCCL with a 1-rank group: the event-driven overlapped all-reduce (comm stream, executor events) against the plain step:
#     same seeds -> the loss after 4 steps must agree with the non-distributed run.
# (2) 2 ranks sharing cuda:0 over gloo (RCCL needs distinct GPUs): rendezvous, broadcast, bucket ranges, MAX-over-ranks timing.
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { python bench.py --steps 4 --warmup 0 --no-profile --no-eval --no-cpu-baseline 2>gpurun_out/dist_$1.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['loss'])"; }
run plain
TINYFACES_FORCE_DIST=1 run rccl1 || tail -5 gpurun_out/dist_rccl1.err
export TINYFACES_BENCH_SHARE_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-profile > gpurun_out/bench_dist2.json 2> gpurun_out/bench_dist2.err
echo "dist exit $?"; tail -3 gpurun_out/bench_dist2.err; cut -c1-400 gpurun_out/bench_dist2.json
