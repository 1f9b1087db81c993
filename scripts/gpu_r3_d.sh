#!/bin/bash
# round 3, call D: quick wins (elementwise prefetch, conv3x3h 112 KiB, wgrad ring depth knob): parity of the touched kernels + A/B
mkdir -p gpurun_out/r3d
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "bn_fused or conv3x3h or wgrad" 2>&1 | tail -5
export STEPS=20
bash scripts/gpu_ab.sh "X=0" "TINYFACES_WGRAD_NS=2" 2>&1 | tee gpurun_out/r3d/ab.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trace_out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_out -o train -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-eval --no-profile --no-fp32-path > $GRAFT_REPO_ROOT/gpurun_out/r3d/trace.log 2>&1
f=$(find /tmp/trace_out -name "*kernel_trace.csv" | head -1)
cd $GRAFT_REPO_ROOT && python scripts/trace_timeline.py "$f" > gpurun_out/r3d/timeline.txt 2>&1
head -12 gpurun_out/r3d/timeline.txt; grep -A12 "main-queue kernel totals" gpurun_out/r3d/timeline.txt
