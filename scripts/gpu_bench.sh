#!/bin/bash
# Run on the GPU box: bench line + rocprofv3 kernel stats of the same command.
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?"; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
if [ -z "$NO_PROF" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o train -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-eval --no-profile > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1
  echo "rocprof exit $?"
  cd $GRAFT_REPO_ROOT
  find gpurun_out/prof -name "*stats*" | head
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -40 "$f"
fi
