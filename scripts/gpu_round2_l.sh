#!/bin/bash
# round 2, call L: halo-resident 3x3 kernel (conv3x3h): correctness, traced micro-benchmark vs the im2col tiles, A/B on the step
mkdir -p gpurun_out gpurun_out/l
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider --timeout 300 -k "conv3x3h" > gpurun_out/l/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/l/pytest.log
tail -25 gpurun_out/l/pytest.log
TAG=r2l PAT="conv_dma|conv3x3h" TILES=13,46,50 LAYERS=l2.c2,l3.c2 bash scripts/gpu_ubench_trace.sh scripts/microbench_r2.py
tail -8 gpurun_out/ub/r2l.log
cd $GRAFT_REPO_ROOT
for v in on off on off; do
  if [ $v = off ]; then export TINYFACES_CONV3H_OFF=1; else unset TINYFACES_CONV3H_OFF; fi
  timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-eval > gpurun_out/l/bench_$v.json 2> gpurun_out/l/bench_$v.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/l/bench_$v.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('forward_pass'))"
done
