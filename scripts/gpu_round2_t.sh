#!/bin/bash
# round 2, call T: conv3x3h with early prologue DMAs + two-tile epilogue: tests, trace, microbench, A/B of the late DMA issue; the two tests that failed in call S
mkdir -p gpurun_out gpurun_out/t
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_conv.py tests/test_gpu_scripts.py tests/test_gpu_fullsize.py -q -m gpu -p no:cacheprovider --timeout 600 -k "conv3x3h or scripts or train_step" > gpurun_out/t/pytest.log 2>&1
echo "pytest exit $?: $(tail -1 gpurun_out/t/pytest.log)"
for late in 0 1; do
  export TINYFACES_CONV3H_LATE=$late
  timeout 200 python scripts/trace_conv3x3h.py > gpurun_out/t/trace_l3_late$late.txt 2>&1; head -3 gpurun_out/t/trace_l3_late$late.txt | cut -c1-330
  TAG=r2t_late$late PAT="conv3x3h" TILES=50 LAYERS=l3.c2 REPS=10 bash scripts/gpu_ubench_trace.sh scripts/microbench_r2.py
  cd $GRAFT_REPO_ROOT
done
for late in 0 1 0 1; do
  export TINYFACES_CONV3H_LATE=$late
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-eval > gpurun_out/t/bench_$late.json 2> gpurun_out/t/bench_$late.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/t/bench_$late.json').read().strip().splitlines()[-1]); print('late $late', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('forward_pass')['ms'])"
done
