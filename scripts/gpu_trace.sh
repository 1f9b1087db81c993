#!/bin/bash
# rocprofv3 kernel trace (timestamps) of 3 training steps -> gpurun_out/trace/train_kernel_trace.csv (gap / overlap analysis)
mkdir -p gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trace_out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_out -o train -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-eval --no-profile ${BENCH_ARGS} > $GRAFT_REPO_ROOT/gpurun_out/trace/run.log 2>&1
echo "rocprof exit $?"
find /tmp/trace_out -name "*kernel_trace.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/trace/ \;
ls -la $GRAFT_REPO_ROOT/gpurun_out/trace
