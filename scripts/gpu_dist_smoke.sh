#!/bin/bash
# Functional checks of the data-parallel path on a 1-GPU box.
# (1) RCCL with a 1-rank group: the event-driven overlapped all-reduce (comm stream, executor events) against the plain step:
#     same seeds -> the loss after 4 steps must agree with the non-distributed run.
# (2) 2 ranks sharing cuda:0 over gloo (RCCL needs distinct GPUs): rendezvous, broadcast, bucket ranges, MAX-over-ranks timing.
# (3) 2 RCCL ranks on the ONE device of the box: does RCCL permit it?  (expected: no -- recorded verbatim in gpurun_out/dist_smoke.txt)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { python bench.py --steps 4 --warmup 0 --no-profile --no-eval --no-cpu-baseline --no-fp32-path 2>gpurun_out/dist_$1.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['loss'])"; }
run plain
TINYFACES_FORCE_DIST=1 run rccl1 || tail -5 gpurun_out/dist_rccl1.err
# (1b) r4: the NATIVE exchange (tf_comm_allreduce_hook: ncclAllReduce issued from C inside the backward enqueue) with the same 1-rank group
TINYFACES_FORCE_DIST=1 TINYFACES_ALLREDUCE_NATIVE=1 run rccl1_native || tail -5 gpurun_out/dist_rccl1_native.err
export TINYFACES_BENCH_SHARE_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-profile > gpurun_out/bench_dist2.json 2> gpurun_out/bench_dist2.err
echo "dist exit $?"; tail -3 gpurun_out/bench_dist2.err; cut -c1-400 gpurun_out/bench_dist2.json

# (3) two nccl ranks, one device
cat > /tmp/nccl2.py <<'PY'
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl")
    t = torch.ones(4, device="cuda:0")
    dist.all_reduce(t); torch.cuda.synchronize()
    print(f"rank {dist.get_rank()}: 2 RCCL ranks on one device WORK, all_reduce(ones) = {t[0].item()}")
except Exception as e:
    print(f"rank {os.environ.get('RANK')}: RCCL refuses two ranks on one device: {type(e).__name__}: {str(e)[:300]}")
PY
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 /tmp/nccl2.py 2>&1 | grep -i "rank \|duplicate\|invalid" | head -6 | tee gpurun_out/dist_smoke_nccl2.txt
