#!/bin/bash
mkdir -p gpurun_out/nms
timeout 900 python -m pytest tests/test_gpu_small_ops.py tests/test_gpu_hard.py -x -q -m gpu -k "nms" 2>&1 | tail -5
bash scripts/gpu_nms.sh
