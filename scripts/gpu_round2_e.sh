#!/bin/bash
# round 2, call E: timeline of a training step (which stream bounds it), dist test re-run, eval-shape tiles
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
bash scripts/gpu_trace.sh > gpurun_out/trace_e.log 2>&1
f=$(ls gpurun_out/trace/*kernel_trace.csv | head -1)
python scripts/trace_timeline.py $f | tee gpurun_out/timeline_e.txt
head -1 $f > gpurun_out/trace/head.txt
rm -f gpurun_out/trace/*kernel_trace.csv
timeout 600 python -m pytest tests/test_gpu_dist.py tests/test_gpu_small_ops.py -q -m gpu -p no:cacheprovider --timeout 600 -k "two_rank or pairwise" > gpurun_out/pytest_e.log 2>&1
tail -5 gpurun_out/pytest_e.log
TAG=r2e_eval PAT="conv_dma" TILES=13,44,45,46,14 bash scripts/gpu_ubench_trace.sh scripts/microbench_eval.py
