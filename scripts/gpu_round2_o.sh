#!/bin/bash
# round 2, call O: conv3x3h with the unrolled-tap K loop: correctness, stage timeline, true kernel durations, step A/B
mkdir -p gpurun_out gpurun_out/o
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider --timeout 300 -k "conv3x3h" > gpurun_out/o/pytest.log 2>&1
echo "pytest exit $?: $(tail -1 gpurun_out/o/pytest.log)"
timeout 200 python scripts/trace_conv3x3h.py > gpurun_out/o/trace_l3.txt 2>&1; head -3 gpurun_out/o/trace_l3.txt
TAG=r2o PAT="conv_dma|conv3x3h" TILES=13,50 LAYERS=l2.c2,l3.c2 REPS=10 bash scripts/gpu_ubench_trace.sh scripts/microbench_r2.py
cd $GRAFT_REPO_ROOT
for v in on off; do
  if [ $v = off ]; then export TINYFACES_CONV3H_OFF=1; else unset TINYFACES_CONV3H_OFF; fi
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-eval > gpurun_out/o/bench_$v.json 2> gpurun_out/o/bench_$v.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/o/bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('forward_pass'))"
done
