#!/bin/bash
# round 2, call B: correctness of the 32x32x16 tiles + traced micro-benchmark of every tile on the layer shapes
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider --timeout 600 -k "mma32 or fullsize" > gpurun_out/pytest_b.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_b.log
tail -15 gpurun_out/pytest_b.log
TAG=r2b PAT="conv_dma" bash scripts/gpu_ubench_trace.sh scripts/microbench_r2.py
cat gpurun_out/ub/r2b.log | tail -40
