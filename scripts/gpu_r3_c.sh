#!/bin/bash
# round 3, call C: what does the data-gradient chain cost when it owns the GPU (no weight gradients at all)?
mkdir -p gpurun_out/r3c
export STEPS=20
bash scripts/gpu_ab.sh "X=0" "TINYFACES_DBG_SKIP_WGRAD=1" "TINYFACES_SINGLE_STREAM=1" 2>&1 | tee gpurun_out/r3c/ab.txt
for cfg in "nowgrad:TINYFACES_DBG_SKIP_WGRAD=1" "single:TINYFACES_SINGLE_STREAM=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trace_out
  env $envs timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_out -o train -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-eval --no-profile --no-fp32-path > $GRAFT_REPO_ROOT/gpurun_out/r3c/trace_$name.log 2>&1
  f=$(find /tmp/trace_out -name "*kernel_trace.csv" | head -1)
  cd $GRAFT_REPO_ROOT && python scripts/trace_timeline.py "$f" > gpurun_out/r3c/timeline_$name.txt 2>&1
  head -12 gpurun_out/r3c/timeline_$name.txt
done
