#!/bin/bash
# round 3, call B: is LDS co-residence what the two streams fight over IN the step?  Sparse weight-gradient grids via the existing knobs.
mkdir -p gpurun_out/r3b
export STEPS=20
bash scripts/gpu_ab.sh "X=0" "TINYFACES_WGRAD_BLOCKS=256" "TINYFACES_WGRAD_BLOCKS=128" "TINYFACES_WGRAD3_BLOCKS=128" "TINYFACES_WGRAD_BLOCKS=256 TINYFACES_WGRAD3_BLOCKS=128" "TINYFACES_WGRAD_BLOCKS=128 TINYFACES_WGRAD3_BLOCKS=128" "TINYFACES_WGRAD_BLOCKS=128 TINYFACES_WGRAD3_BLOCKS=64" 2>&1 | tee gpurun_out/r3b/ab.txt
for cfg in "base:X=0" "sparse:TINYFACES_WGRAD_BLOCKS=128 TINYFACES_WGRAD3_BLOCKS=128"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trace_out
  env $envs timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_out -o train -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-eval --no-profile > $GRAFT_REPO_ROOT/gpurun_out/r3b/trace_$name.log 2>&1
  f=$(find /tmp/trace_out -name "*kernel_trace.csv" | head -1)
  cd $GRAFT_REPO_ROOT && python scripts/trace_timeline.py "$f" > gpurun_out/r3b/timeline_$name.txt 2>&1
  head -12 gpurun_out/r3b/timeline_$name.txt
done
