#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command (CSV), copied to gpurun_out/prof_csv
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -o train -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-eval --no-profile ${BENCH_ARGS} > $GRAFT_REPO_ROOT/gpurun_out/prof_run.log 2>&1
echo "rocprof exit $?"
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_csv
find /tmp/prof_out -name "*stats*.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_csv/ \;
ls $GRAFT_REPO_ROOT/gpurun_out/prof_csv
head -30 $GRAFT_REPO_ROOT/gpurun_out/prof_csv/*kernel_stats.csv
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof_run.log
