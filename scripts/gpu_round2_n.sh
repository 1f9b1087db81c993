#!/bin/bash
# round 2, call N: the three K-loop variants of conv3x3h: correctness, stage timeline, true kernel durations
mkdir -p gpurun_out gpurun_out/n
export PYTHONUNBUFFERED=1
for v in 0 1 2; do
  export TINYFACES_CONV3H_VAR=$v
  timeout 300 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider --timeout 300 -k "conv3x3h" > gpurun_out/n/pytest_v$v.log 2>&1
  echo "VAR $v pytest exit $?: $(tail -1 gpurun_out/n/pytest_v$v.log)"
  timeout 200 python scripts/trace_conv3x3h.py > gpurun_out/n/trace_l3_v$v.txt 2>&1; head -3 gpurun_out/n/trace_l3_v$v.txt
  TAG=r2n_v$v PAT="conv3x3h" TILES=50 LAYERS=l2.c2,l3.c2 REPS=10 bash scripts/gpu_ubench_trace.sh scripts/microbench_r2.py
  cd $GRAFT_REPO_ROOT
done
