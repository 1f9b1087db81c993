#!/bin/bash
# round 2, call M: stage timeline of conv3x3h (s_memtime stamps) + ablations (no steady-state DMA / no MFMA) under the kernel trace
mkdir -p gpurun_out gpurun_out/m
export PYTHONUNBUFFERED=1
timeout 200 python scripts/trace_conv3x3h.py > gpurun_out/m/trace_l3.txt 2>&1; cat gpurun_out/m/trace_l3.txt
SHAPE=12,63,63,128,128 timeout 200 python scripts/trace_conv3x3h.py > gpurun_out/m/trace_l2.txt 2>&1; head -9 gpurun_out/m/trace_l2.txt
for d in 0 1 2 3; do
  export TINYFACES_CONV3H_DBG=$d
  TAG=r2m_dbg$d PAT="conv3x3h" TILES=50 LAYERS=l2.c2,l3.c2 REPS=6 bash scripts/gpu_ubench_trace.sh scripts/microbench_r2.py
  cd $GRAFT_REPO_ROOT
done
