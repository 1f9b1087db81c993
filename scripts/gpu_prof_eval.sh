#!/bin/bash
# rocprofv3 kernel-trace stats of the eval pyramid leg (configs[1]) -> gpurun_out/prof_eval_csv
mkdir -p gpurun_out
python bench.py --eval-only > gpurun_out/eval_only.json 2> gpurun_out/eval_only.err; cat gpurun_out/eval_only.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_eval
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_eval -o eval -- python $GRAFT_REPO_ROOT/bench.py --eval-only > $GRAFT_REPO_ROOT/gpurun_out/prof_eval_run.log 2>&1
echo "rocprof exit $?"
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof_eval_csv
find /tmp/prof_eval -name "*stats*.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/prof_eval_csv/ \;
head -40 $GRAFT_REPO_ROOT/gpurun_out/prof_eval_csv/*kernel_stats.csv
