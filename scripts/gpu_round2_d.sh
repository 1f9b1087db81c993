#!/bin/bash
# round 2, call D: all-taps 3x3 weight gradient: parity, traced micro-benchmark, A/B of the training step on ONE box
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_dist.py -q -m gpu -p no:cacheprovider --timeout 600 -k "wgrad or fullsize_train or train_step or real_layer or trainer or engine or dual_stream or slice or two_rank" > gpurun_out/pytest_d.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_d.log
tail -12 gpurun_out/pytest_d.log
TAG=r2d PAT="wgrad" bash scripts/gpu_ubench_trace.sh scripts/microbench_wgrad3.py
grep -v "^W2026\|^E2026\|rocprof" gpurun_out/ub/r2d.log | tail -8
for rep in 1 2; do
  for off in 1 0; do
    if [ $off = 1 ]; then export TINYFACES_WGRAD3_OFF=1; else unset TINYFACES_WGRAD3_OFF; fi
    timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-eval > gpurun_out/bench_d_off${off}_${rep}.json 2> gpurun_out/bench_d_off${off}_${rep}.err
    python - <<PY
import json
d=json.load(open("gpurun_out/bench_d_off${off}_${rep}.json"))
print("WGRAD3_OFF=${off} rep ${rep}:", d["value"], "img/s", d["ms_per_step"], "ms", [ (k["kernel"], k["ms_per_step"]) for k in d.get("kernels", [])], d.get("roofline",{}).get("forward_pass"), d.get("kernels_single_stream",{}).get("ms_per_step"))
PY
  done
done
