#!/bin/bash
# usage: gpu_ubench_trace.sh <python script + args> ; env TAG names the output; true kernel durations via rocprofv3 kernel trace
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/ub
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ub_$TAG
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ub_$TAG -o ub -- python $GRAFT_REPO_ROOT/$@ > $GRAFT_REPO_ROOT/gpurun_out/ub/$TAG.log 2>&1
f=$(find /tmp/ub_$TAG -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/scripts/trace_summary.py $f "${PAT:-conv_dma|wgrad}" | tee $GRAFT_REPO_ROOT/gpurun_out/ub/$TAG.txt
